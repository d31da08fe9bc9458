"""Launch planning of the inference chain: which plain layers run in Winograd form, which layers hand their activation over in split
form, which fuse their ToRGB.  Pure functions of (batch, layer shapes, functional.Config), cached per generator.  Mixed into
model.Generator."""
from . import functional as F_


class PlannerMixin:
    def _wino_inputs(self, batch, layers):
        """{index of a plain layer the inference chain runs in 1-D Winograd form: outputs per tile (2 | 4)} (functional.
        wsplit_chain_f: fed by a transposed conv + blur that can hand over the transformed input, enough input channels for the
        smaller MFMA count to outweigh the larger hand-over).  A pure function of (batch, layer shapes, switches): the range plan
        and the launch plan both ask it."""
        key = (batch, F_.config())
        cache = self.__dict__.setdefault('_wino_cache', {})
        if key not in cache:
            out, res = {}, self.input.input.shape[2]
            for li, layer in enumerate(layers):
                c = layer.conv
                if c.upsample:
                    res *= 2
                elif li >= 2 and layers[li - 1].conv.upsample and F_.rgb_fusable(batch, c.in_channel, c.out_channel, res, res):
                    f = F_.wsplit_chain_f(batch, c.in_channel, c.out_channel, res, res)
                    if f:
                        out[li] = f
            cache[key] = out
        return cache[key]

    def _chain_plan(self, batch, chain, noise, layers):
        """Per layer [runs on the split chain, fuses its ToRGB, hands its output over in split form, stores y, Winograd form of the
        hand-over (0 | 2 | 4), arithmetic of that hand-over (None | 'fp16f8'), arithmetic of the PLAIN split hand-over of an F(4,3)
        layer to the transposed conv after it (None | 'fp16f8')].  Depends only on (batch, configuration, which noises are given): cached, so a forward does not
        query the library 40 times (it matters for the launch-bound small batches)."""
        key = (batch, chain, F_.config(), tuple(n is None for n in noise))
        plan = self._chain_plans.get(key) if hasattr(self, '_chain_plans') else None
        if plan is None:
            plan, res, res_in = [], self.input.input.shape[2], []
            for li, layer in enumerate(layers):
                c = layer.conv
                up = c.upsample
                res_out = 2 * res if up else res
                res_in.append(res)
                nxt = layers[li + 1].conv if li + 1 < len(layers) else None
                mode = F_.N.MODE_UP3 if up else F_.N.MODE_PLAIN3
                use_chain = chain and noise[li] is not None and F_.split_ok(batch, c.in_channel, c.out_channel, res, res, mode)
                fuse = use_chain and (not up) and F_.rgb_fusable(batch, c.in_channel, c.out_channel, res, res)
                # hand the activation to the next conv in its own split input form when it can stage that by DMA (and,
                # for a plain conv, when nothing else needs the fp32 tensor: its ToRGB is fused into this launch)
                to_next = use_chain and nxt is not None and (up or fuse) and nxt.kernel_size == 3 and \
                    noise[li + 1] is not None and \
                    F_.xin_ok(batch, nxt.in_channel, nxt.out_channel, res_out, res_out,
                              F_.N.MODE_UP3 if nxt.upsample else F_.N.MODE_PLAIN3)
                want_y = not (fuse and (nxt is None or to_next))
                plan.append([use_chain, fuse, to_next, want_y, 0, None, None])
                res = res_out
            # plain layers in Winograd form: the producing transposed conv's blur hands over the transformed input
            for li, f in (self._wino_inputs(batch, layers).items() if chain else ()):
                if plan[li][0] and plan[li][1] and plan[li - 1][0] and plan[li - 1][2]:
                    plan[li - 1][4] = f
                    c = layers[li].conv
                    plan[li - 1][5] = F_.wsplit_chain_arith(batch, c.in_channel, c.out_channel, res_in[li], res_in[li], f)
                    nxt = layers[li + 1].conv if li + 1 < len(layers) else None
                    if f == 4 and plan[li][2] and nxt is not None and nxt.upsample and plan[li + 1][0]:
                        plan[li][6] = F_.xs_chain_arith(batch, c.in_channel, c.out_channel, res_in[li], res_in[li], nxt.out_channel)
            # direct plain layers fed by a transposed conv + blur in plain split form: fp8 cross-term operands where the consumer's
            # plan reads them (Config.cross_terms)
            for li in range(1, len(layers)):
                c = layers[li].conv
                if chain and not c.upsample and layers[li - 1].conv.upsample and plan[li][0] and plan[li][1] and plan[li - 1][0] and \
                        plan[li - 1][2] and plan[li - 1][4] == 0:
                    plan[li - 1][5] = F_.xs_plain_arith(batch, c.in_channel, c.out_channel, res_in[li], res_in[li])
            if not hasattr(self, '_chain_plans'):
                self._chain_plans = {}
            self._chain_plans[key] = plan
        return plan
