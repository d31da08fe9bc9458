"""Range plan of the fp16-split arithmetic and the generator's saturation word (what keeps `fp16x3` fp32-grade):
calibration of every conv input's magnitude once per weight version, a RangeToken per forward, the non-blocking poll, the
fallback to `bf16x3`.  Mixed into model.Generator; nothing here launches a conv itself."""
import math
import struct
import warnings

import torch

from . import functional as F_


class RangeToken:
    """One no-grad fp16x3 forward's claim on its generator's saturation word: `snap` (pinned host int32) receives the word's
    value right after the forward's last launch, `event` marks that copy; `delta` = pairs clamped by THIS forward, filled in
    when the token is checked (Generator.range_ok / the non-blocking poll of the next forward).  `stamp` = the weights the forward
    ran on (a saturation of OLD weights must not switch the arithmetic of new ones); `suspect`: another forward that was in flight
    beside this one (other HIP stream) clamped operands and the shared word cannot tell the two apart -- range_ok() says False."""
    __slots__ = ('event', 'snap', 'delta', 'stamp', 'suspect', 'version')

    def __init__(self, event, snap, stamp=None, version=0):
        self.event, self.snap, self.delta, self.stamp, self.suspect = event, snap, None, stamp, False
        self.version = version      # the range plan's version the forward ran under: a clamp under a plan that has since been
                                    # widened is no evidence against the widened one (range_ok re-renders, but does not fall back)


_PINNED_WORDS = []      # free list of pinned int32 [1] host tensors (a fresh pin_memory() per forward would cost ~20 us)


def _pinned_word():
    return _PINNED_WORDS.pop() if _PINNED_WORDS else torch.zeros(1, dtype=torch.int32).pin_memory()


class RangePlanMixin:
    """Methods of model.Generator (which provides input, convs, noises, _weights_stamp, _wino_inputs, _drop_graphs)."""
    MAX_PENDING_TOKENS = 8          # unchecked forwards in flight before the oldest token is awaited (host far ahead of the device)
    # A plan is calibrated on the first batch after a weight change; a later, systematically "louder" stream would otherwise pay
    # a bf16x3 re-render per batch for good.  After a fallback the NEXT no-grad forward measures its own batch (every row) and
    # WIDENS the plan (element-wise max with the old one), returning the generator to fp16x3 -- at most this many times per
    # weight version (then bf16x3 stays: 1.1e-4 instead of 1.4e-5 from the fp64 evaluation, still inside the 1e-3 contract).
    AUTO_RECALIBRATIONS = 3
    CALIBRATION_ROWS = 8            # rows of the first batch the initial calibration looks at (a recalibration takes every row,
    CALIBRATION_CHUNK = 64          #  this many at a time)

    def _sat_word(self):
        """This generator's saturation word (functional.saturation_sink): one int32 on the weights' device, owned by the
        instance -- not a buffer (never in the state_dict), deep-copied with the module, re-made after .to(device)."""
        dev = self.input.input.device
        w = self.__dict__.get('_sat')
        if w is None or w.device != dev:
            # tokens of the OLD word can never be resolved against the new one: whoever still holds one must re-render
            for tok in self.__dict__.get('_sat_tokens') or []:
                tok.delta, tok.suspect = 0, True
                if tok.snap is not None:
                    tok.event.synchronize()             # (the async copy into the pinned word must have landed before it is reused)
                    _PINNED_WORDS.append(tok.snap)
                    tok.snap = None
            w = self.__dict__['_sat'] = F_.new_saturation_word(dev)
            self.__dict__['_sat_seen'] = 0
            self.__dict__['_sat_tokens'] = []
        return w

    def _snapshot(self):
        """Queue an async copy of the saturation word into pinned host memory behind everything launched so far on the current
        stream; returns the RangeToken (None while capturing a graph)."""
        if torch.cuda.is_current_stream_capturing():
            return None
        word = self._sat_word()
        snap = _pinned_word()
        snap.copy_(word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st = getattr(self, '_range_state', None)
        tok = RangeToken(ev, snap, st['stamp'] if st is not None else None, st.get('version', 0) if st is not None else 0)
        self._sat_tokens.append(tok)
        return tok

    def _check_tokens(self, upto=None):
        """Resolve queued tokens in order: all that are complete (upto=None, never blocks) or everything up to and including
        `upto` (blocks on its event).  Returns the number of newly seen saturated pairs."""
        toks = self.__dict__.get('_sat_tokens')
        new = 0
        while toks:
            tok = toks[0]
            if upto is not None:
                tok.event.synchronize()
            elif not tok.event.query():
                break
            toks.pop(0)
            val = int(tok.snap[0]) & 0xffffffff
            # Snapshots of forwards on DIFFERENT streams (functional.StreamPipeline) may be taken out of queue order: a later
            # token can hold a smaller value.  The running maximum (as a signed 32-bit distance: the word may wrap) keeps the
            # deltas non-negative; whoever was still in flight when a non-zero delta shows up is marked suspect, because the
            # shared word cannot say which of the overlapping forwards clamped.
            dist = (val - self._sat_seen) & 0xffffffff
            tok.delta = dist if dist < 0x80000000 else 0
            if tok.delta:
                self.__dict__['_sat_seen'] = val
                for other in toks:
                    other.suspect = True
            new += tok.delta                            # (a merely suspect token -- delta 0 -- is re-rendered by whoever asks
                                                        #  range_ok(), but it is no evidence against the plan)
            _PINNED_WORDS.append(tok.snap)
            tok.snap = None
            if tok is upto:
                break
        return new

    def _settle_oldest_if_full(self):
        """Every fp16x3 forward leaves a token -- no forward goes unchecked.  When MAX_PENDING_TOKENS are already queued (the host
        runs that far ahead of the device) the OLDEST one is awaited first: the host then trails the device by at most that
        many forwards, and a saturation in the awaited forward switches the arithmetic at once."""
        toks = self.__dict__.get('_sat_tokens')
        if toks and len(toks) >= self.MAX_PENDING_TOKENS and not torch.cuda.is_current_stream_capturing():
            seen = self._check_tokens(upto=toks[0])
            if seen:
                self._fall_back(seen, 'in earlier forwards')

    def saturated_pairs(self):
        """fp16 operand pairs this generator's launches (forward and backward) clamped or found non-finite so far
        (synchronises the device).  0 = the fp32-grade claim of the fp16x3 arithmetic held for everything it produced."""
        return int(self._sat_word().item()) & 0xffffffff

    def _fall_back(self, pairs, where):
        st = getattr(self, '_range_state', None)
        if st is not None and st['mode'] == 'fp16x3':
            st['mode'] = 'bf16x3'
            again = st.get('recal_left', self.AUTO_RECALIBRATIONS) > 0
            st['recal'] = again                         # the next no-grad forward re-measures its batch and widens the plan
            warnings.warn('Generator: %d fp16 operand pairs left the planned range (or were NaN/Inf) %s; this generator runs the '
                          'bf16x3 arithmetic (fp32 exponent range) %s' % (pairs, where, 'until its next forward has widened the '
                          'range plan' if again else 'until its weights change or recalibrate_ranges() is called'),
                          RuntimeWarning, stacklevel=4)

    def range_ok(self, token):
        """Did the forward behind `token` stay inside the fp16 range plan?  Blocks until that forward has finished (and only that
        far).  False: it clamped operands -- the generator has switched itself to bf16x3, re-render the batch."""
        if token is None:
            return True
        if token.delta is None:
            self._check_tokens(upto=token)
        if token.delta is None:
            # not in the queue any more (the word was re-made on another device, or the queue was reset): nothing can vouch
            # for this forward -- unverifiable reads as "re-render", never as "verified"
            token.delta, token.suspect = 0, True
        if token.delta:
            st = getattr(self, '_range_state', None)
            # (a token of weights that have since been replaced: re-render, but leave the NEW weights' calibrated plan alone)
            # (nor a token of a plan that has been widened since: the other chunk of a StreamPipeline would otherwise spend a second
            #  widening credit and drop the graphs on evidence that belongs to the old plan)
            if st is not None and (token.stamp is None or token.stamp == st['stamp']) and token.version == st.get('version', 0):
                self._fall_back(token.delta, 'in the forward just checked')
            return False
        return not token.suspect

    def range_mode(self):
        """Arithmetic the next no-grad forward of this generator will run in ('fp16x3' with a live range plan, its fallback, or
        functional.PRECISION when no plan applies)."""
        st = getattr(self, '_range_state', None)
        cfg = getattr(self, 'config', None) or F_.config()
        if cfg.precision == 'fp16x3' and cfg.range_plan is True and st is not None:
            return st['mode']
        return cfg.precision

    def range_stats(self):
        """Counters of this generator's fp16x3 range plan since its weights last changed: verified forwards that were rendered a
        second time (`rerendered`: each one costs that batch a second forward), plan widenings (`widenings`), the arithmetic the
        next forward runs in (`mode`), and what the plan allows per layer (`x_log2`)."""
        st = getattr(self, '_range_state', None) or {}
        return {'rerendered': self.__dict__.get('_rerendered', 0), 'widenings': st.get('version', 0), 'mode': self.range_mode(),
                'x_log2': list(st.get('x_log2', []))}

    def take_range_token(self):
        """The RangeToken of the latest no-grad forward (None when that forward did not run in fp16x3)."""
        tok = self.__dict__.get('_last_token')
        self.__dict__['_last_token'] = None
        return tok

    def recalibrate_ranges(self, styles=None, widen=True, **forward_kwargs):
        """Re-measure the activation ranges of the fp16x3 plan on a batch of the caller's choosing -- for a stream whose later
        batches are louder than the one the plan was calibrated on (the first after a weight change).  With `styles` (and the
        forward's keyword arguments: input_is_latent, truncation, truncation_latent ...) that batch is rendered now, every row
        measured; without, the next no-grad forward measures its own batch.  widen=True keeps every layer's bound at least as
        large as before (a plan that also covers the earlier batches); False replaces it.  The generator is back in fp16x3
        afterwards, with a fresh budget of automatic widenings.  Returns the forward's result when `styles` was given."""
        st = getattr(self, '_range_state', None)
        if st is not None:
            st['recal'], st['recal_widen'], st['recal_left'] = True, bool(widen), self.AUTO_RECALIBRATIONS + 1
        self._drop_graphs()
        if styles is None:
            return None
        with torch.no_grad():
            return self.forward(styles, graph=False, **forward_kwargs)

    def _calibrate_ranges(self, latent, noise, specs, layers, max_rows=None):
        """One forward of (at most CALIBRATION_ROWS rows of) this batch on the fp32 kernels, recording max |x| of every 3x3
        conv's input: x_log2[l] = floor(log2 max)+1 feeds the range plan of functional.styles_batched.  Runs once per weight
        version (tracked like the weight packs; after `.data` edits call invalidate_packs()) and once per widening
        (recalibrate_ranges / after a fallback).  One device->host read."""
        rows = min(latent.shape[0], max_rows or self.CALIBRATION_ROWS)
        chunks = []
        sd_of_layer = [0] + [2 + 3 * (i // 2) + (i % 2) for i in range(len(self.convs))]
        for r0 in range(0, rows, self.CALIBRATION_CHUNK):          # (a B > 64 batch is measured 64 rows at a time: fp32 intermediates)
            n = min(self.CALIBRATION_CHUNK, rows - r0)
            lat = latent[r0:r0 + n].contiguous()
            words = torch.zeros(len(layers), device=latent.device, dtype=torch.int32)
            with F_.precision('fp32'):
                sd = F_.styles_batched(lat, specs)
                x = self.input.input
                for li, layer in enumerate(layers):
                    F_.absmax(x, per_image=False, out=words[li:li + 1])
                    nz = noise[li]
                    if nz is not None and nz.shape[0] != 1:
                        nz = nz[r0:r0 + n]
                    x = layer(x, None, noise=nz, batch=n if li == 0 else None, sd=sd[sd_of_layer[li]], ranged=True)
            chunks.append(words)
        # (max of non-negative floats == max of their bit patterns as int32; NaN/Inf patterns sort above every finite one)
        bits = torch.stack(chunks).max(0).values.cpu().tolist()
        x_log2, bad = [], False
        for b in bits:
            v = struct.unpack('f', struct.pack('I', b & 0xffffffff))[0]
            if not math.isfinite(v):
                bad = True
                x_log2.append(F_.DESIGN_X_LOG2)
            else:
                x_log2.append(0 if v == 0.0 else int(math.floor(math.log2(v))) + 1)
        return x_log2, bad

    def _range_plans(self, latent, noise, specs, order, layers):
        """(plans for styles_batched, arithmetic to run this forward in).  fp16x3 only."""
        st = getattr(self, '_range_state', None)
        stamp = self._weights_stamp()
        capturing = torch.cuda.is_current_stream_capturing()
        self._sat_word()
        if st is None or st['stamp'] != stamp:
            if capturing:
                raise RuntimeError('Generator: the first forward after a weight change calibrates activation ranges (one host '
                                   'read) and cannot run inside a graph capture: run one forward before capturing')
            x_log2, bad = self._calibrate_ranges(latent, noise, specs, layers)
            if self._sat_tokens:                        # forwards of the previous weights: settle them (the calibration synced anyway)
                self._check_tokens(upto=self._sat_tokens[-1])
            st = self._range_state = {'stamp': stamp, 'x_log2': x_log2, 'mode': 'fp16x3'}
            if bad:
                st['mode'] = 'fp32'
                warnings.warn('Generator: non-finite activations during range calibration; this generator runs on the fp32 '
                              'kernels until its weights change', RuntimeWarning, stacklevel=3)
        elif st.get('recal') and not capturing:
            # a fallback (or recalibrate_ranges) asked for it: measure THIS batch, every row, and widen / replace the plan
            x_log2, bad = self._calibrate_ranges(latent, noise, specs, layers, max_rows=latent.shape[0])
            if self._sat_tokens:
                self._check_tokens(upto=self._sat_tokens[-1])
            st['recal'] = False
            st['version'] = st.get('version', 0) + 1        # tokens taken under the old plan stop counting against this one
            st['recal_left'] = st.get('recal_left', self.AUTO_RECALIBRATIONS) - 1
            if bad:
                st['mode'] = 'bf16x3'                  # non-finite activations in this batch: no plan can hold them
            else:
                st['x_log2'] = [max(a, b) for a, b in zip(st['x_log2'], x_log2)] if st.get('recal_widen', True) else x_log2
                st['mode'] = 'fp16x3'
            st['recal_widen'] = True
            self._drop_graphs()                         # captured launch sequences carry the old plan's scales
        elif st['mode'] == 'fp16x3' and not capturing and self._sat_tokens:
            # every earlier forward left a token; the ones already finished are checked here WITHOUT blocking, so a
            # saturating batch is noticed one or two forwards later even by callers that never ask (verify_range=False)
            seen = self._check_tokens()
            if seen:
                self._fall_back(seen, 'in earlier forwards')
        if st['mode'] != 'fp16x3':
            return None, st['mode']
        conv_layer = {id(l.conv): i for i, l in enumerate(layers)}
        # (layers that take their input in Winograd form: |B^T (x*s)| <= 2 (F(2,3)) / 10 (F(4,3)) max|x*s| -- 1 / 4 more binades)
        wino = self._wino_inputs(latent.shape[0], layers)
        plans = [(st['x_log2'][conv_layer[id(m)]] + F_.WSPLIT_GROWTH_LOG2.get(wino.get(conv_layer[id(m)], 0), 0),
                  F_.CALIBRATION_HEADROOM) if id(m) in conv_layer else None for m, _ in order]
        return plans, 'fp16x3'
