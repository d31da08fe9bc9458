"""hipGraph replay of repeated no-grad forwards: which calls may be replayed (_graph_key), the capture at the third call
of a signature, the replay with its range-token bookkeeping.  Mixed into model.Generator."""
import os

import torch

from . import functional as F_
from . import timing

USE_GRAPHS = os.environ.get('SGDFR_GRAPHS', '1') != '0'      # hipGraph replay of repeated no-grad forwards (Generator.forward)


class GraphReplayMixin:
    """Methods of model.Generator (which provides _forward_impl, _weights_stamp, range_mode, the RangePlanMixin)."""
    GRAPH_AFTER = 2                 # eager no-grad forwards of one signature before the next one is captured as a hipGraph
    # Captured signatures kept per generator.  Each capture keeps EVERY intermediate of its forward alive in a private memory pool
    # (about 70 MB per image at 256^2, cm=1: 4.5 GB for a B=64 capture; the eager path frees layer by layer), and each HIP stream
    # that replays gets its own capture (the stream id is part of the key).  Host-bound forwards (the default policy) are small;
    # big ones are captured when the caller asks for it (graph=True / verify_range=True) or, for calls verified by default, when
    # the capture is a small share of the device's memory (_capture_fits) -- at most MAX_BIG_GRAPHS at a time.
    MAX_GRAPHS = 4
    MAX_BIG_GRAPHS = 2

    def _drop_graphs(self):
        self.__dict__.pop('_graphs', None)
        self.__dict__.pop('_graph_calls', None)

    GRAPH_MAX_WORK = 6              # default policy: replay when batch * (size / 256)^2 <= this (host-bound forwards), or when verified

    def _graph_key(self, styles, return_latents, inject_index, truncation, truncation_latent, input_is_latent, noise,
                   randomize_noise, image_out, verify_range, graph, verify_explicit=False):
        """Signature under which a no-grad forward may be replayed as a hipGraph, or None when it must run eagerly: gradients,
        style mixing, caller-supplied or fresh noise, a caller-owned uint8 target, hooks, an enclosing capture, launch timing --
        and forwards that are not host-bound (GRAPH_MAX_WORK) unless the caller asked for a replay (graph=True) or for a verified
        forward (verify_range=True passed EXPLICITLY: generate_image, whose wait exposes the enqueue time).  A forward that is
        verified only because that is the default (a raw `G([w])`) keeps the gate: no multi-GB capture behind the caller's back."""
        if graph is False or not USE_GRAPHS or not getattr(self, 'use_graphs', True) or torch.is_grad_enabled() or \
                timing.active() is not None:
            return None
        if len(styles) != 1 or inject_index is not None or noise is not None or randomize_noise:
            return None
        w = styles[0]
        if not isinstance(w, torch.Tensor) or not w.is_cuda or w.dtype != torch.float32 or w.requires_grad:
            return None
        if graph is None and not (verify_range and verify_explicit) and self._graph_work(w) > self.GRAPH_MAX_WORK:
            # not host-bound, and nobody asked for a replay.  A forward that is verified by default (the reference-shaped `G([w])`)
            # still exposes its ~1 ms of enqueue time to the wait: it is replayed when the capture's pinned intermediates are a small
            # share of THIS device's memory (round 6: 4.5 GB for B=64 on a 288 GB part), otherwise it keeps running eagerly
            if not (verify_range and self._capture_fits(w)):
                return None
        if image_out is not None and image_out.frames is not None:
            return None
        if truncation < 1 and truncation_latent is None:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        mods = self.__dict__.get('_all_mods')
        if mods is None:
            mods = self.__dict__['_all_mods'] = list(self.modules())
        for m in mods:
            if m._forward_hooks or m._forward_pre_hooks:
                return None
        u8 = None if image_out is None else ('u8', image_out.swap_rb)
        return (tuple(w.shape), bool(input_is_latent), bool(return_latents), float(truncation),
                None if truncation >= 1 else tuple(truncation_latent.shape), u8, F_.config(), bool(self.overlap_rgb), w.device,
                # a graph's static input / output buffers belong to the stream that replays it: forwards queued on different
                # streams (functional.StreamPipeline) get their own capture instead of racing on one
                F_.N.stream().value)

    def _graph_work(self, w):
        return w.shape[0] * (self.size / 256.0) ** 2

    CAPTURE_BYTES_PER_WORK = 72e6       # intermediates a capture pins per unit of _graph_work (256^2 image, cm=1; cm=2: twice)
    CAPTURE_MAX_SHARE = 0.03            # of the device's total memory, per capture (and at most a quarter of what is free now)

    def _capture_fits(self, w):
        """May a big forward be captured WITHOUT the caller asking?  Only when its pinned intermediates are a small share of the
        device: <= CAPTURE_MAX_SHARE of total memory and <= 25 % of the memory free right now."""
        est = self.CAPTURE_BYTES_PER_WORK * self._graph_work(w) * (2 if getattr(self, 'channel_multiplier', 1) >= 2 else 1)
        try:
            free, total = torch.cuda.mem_get_info(w.device)
        except Exception:       # noqa: BLE001
            return False
        return est <= self.CAPTURE_MAX_SHARE * total and est <= 0.25 * free

    def _replay_or_run(self, key, styles, return_latents, return_features, inject_index, truncation, truncation_latent, input_is_latent,
                       image_out, verify_range):
        """forward() for a call that has a graph signature: eager until the signature has settled, then capture once, then replay."""
        graphs = self.__dict__.setdefault('_graphs', {})
        calls = self.__dict__.setdefault('_graph_calls', {})
        stamp, mode = self._weights_stamp(), self.range_mode()
        entry = graphs.get(key)
        if entry is not None and (entry['stamp'] != stamp or entry['mode'] != mode):
            self._drop_graphs()                            # new weights / the generator changed arithmetic: every graph is stale
            graphs = self.__dict__.setdefault('_graphs', {})
            calls = self.__dict__.setdefault('_graph_calls', {})
            entry = None
        w = styles[0]
        trunc = truncation_latent if truncation < 1 else None
        if entry is not None and entry['mode'] == 'fp16x3' and self.__dict__.get('_sat_tokens'):
            # the non-blocking poll of the eager path (_range_plans), which a replay never reaches: tokens of finished forwards are
            # resolved here, and a saturation seen in any of them switches this generator to bf16x3 before the next replay
            seen = self._check_tokens()
            if seen:
                self._fall_back(seen, 'in earlier forwards')
                self._drop_graphs()
                return self._forward_impl(styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                          input_is_latent, None, False, image_out, verify_range)
        if entry is None:
            n = calls.get(key, 0)
            if n < self.GRAPH_AFTER:                       # not yet: packs, launch plans and the range calibration settle eagerly
                calls[key] = n + 1
                return self._forward_impl(styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                          input_is_latent, None, False, image_out, verify_range)
            st = getattr(self, '_range_state', None)
            if F_.config().precision == 'fp16x3' and F_.config().range_plan is True and \
                    (st is None or st['stamp'] != stamp or st.get('recal')):
                return self._forward_impl(styles, return_latents, return_features, inject_index, truncation, truncation_latent,
                                          input_is_latent, None, False, image_out, verify_range)       # calibrate first (host read)
            s_in = w.detach().clone()
            s_tr = trunc.detach().clone() if trunc is not None else None
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            with F_.capture_graph(g):
                out = self._forward_impl([s_in], return_latents, False, None, truncation, s_tr, input_is_latent, None, False,
                                         None if image_out is None else F_.U8Target(None, 0, image_out.swap_rb), False)
            mode = self.range_mode()
            entry = {'graph': g, 'in': s_in, 'trunc': s_tr, 'out': out, 'stamp': stamp, 'mode': mode}
            entry['big'] = self._graph_work(w) > self.GRAPH_MAX_WORK
            while len(graphs) >= self.MAX_GRAPHS:
                graphs.pop(next(iter(graphs)))
            if entry['big']:
                big = [k for k, e in graphs.items() if e.get('big')]
                for k in big[:max(0, len(big) + 1 - self.MAX_BIG_GRAPHS)]:
                    graphs.pop(k)
            graphs[key] = entry
        entry['in'].copy_(w)
        if entry['trunc'] is not None:
            entry['trunc'].copy_(trunc)
        entry['graph'].replay()
        img, lat = entry['out']
        res = (img.clone(), lat.clone() if lat is not None else None)
        self.__dict__['_last_token'] = None
        if entry['mode'] == 'fp16x3' and F_.config().precision == 'fp16x3' and F_.config().range_plan is True:
            # the captured launches add to this generator's saturation word like eager ones: snapshot it behind the replay
            self._settle_oldest_if_full()
            tok = self._snapshot()
            if not verify_range:
                self.__dict__['_last_token'] = tok
            elif not self.range_ok(tok):                   # this batch clamped operands: render it again, eagerly (measures the batch and
                self._drop_graphs()                        # widens the plan while the budget lasts, else bf16x3), verified like the first
                return self._forward_impl(styles, return_latents, return_features, inject_index, truncation,
                                          truncation_latent, input_is_latent, None, False, image_out, True, 1)
        return res
