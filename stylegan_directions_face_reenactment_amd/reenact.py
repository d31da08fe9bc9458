"""Batched reenactment driver: the MI355X counterpart of the per-frame loop of the reference's
``run_inference.py:170-181``.

The reference re-renders ONE target frame per generator call:
    shift = A(shift_vector [1,15]) ; generate_image(G, source_code, 0.7, trunc, w_plus, 8, shift_code=shift,
                                                     input_is_latent=True)
Every frame depends only on the fixed source code and its own shift vector, so N frames are exactly N rows of one
batch.  `ReenactmentSession.frames` chunks the shift vectors, broadcasts the source W+ code inside
``sgdfr_latent_prepare_f32`` (shift add + truncation in the same launch) and yields [b,3,H,W] images, optionally
converted on the GPU to the uint8 HWC layout the reference writes (libs/utilities/image_utils.py:97-110).
"""
import ctypes
import os

import numpy as np
import torch

from . import _native as N
from . import functional as F_


def images_to_uint8(images):
    """[B,3,H,W] fp32 in [-1,1] -> [B,H,W,3] uint8 with the reference's scaling (image_utils.py:87-110)."""
    N.require_device(images)
    x = N.f32c(images)
    B, C, H, W = x.shape
    if C != 3:
        raise RuntimeError('expected RGB images, got %d channels' % C)
    y = torch.empty(B, H, W, 3, device=x.device, dtype=torch.uint8)
    N.call('sgdfr_image_to_u8_f32', N.ptr(x), N.ptr(y), B, H, W, N.stream())
    return y


def grid_frames_uint8(panels, swap_rb=False, out=None):
    """Side-by-side video frames for a batch: panels = list of [B,3,H,W] or [1,3,H,W] (shown in every frame) fp32
    images in [-1,1] -> [B,H,K*W,3] uint8.  One launch for what the reference does per frame with
    generate_grid_image + tensor_to_image + np.uint8 (utils_inference.py:11-33, run_inference.py:188-194);
    swap_rb applies that path's cvtColor channel swap.  A panel given as None is left untouched in `out` (the generator
    already wrote it there: Generator.forward(image_out=U8Target(out, panel)))."""
    if not 1 <= len(panels) <= 4:
        raise RuntimeError('grid_frames_uint8 takes 1..4 panels, got %d' % len(panels))
    xs = []
    for x in panels:
        if x is None:
            xs.append(None)
            continue
        N.require_device(x)
        xs.append(N.f32c(x if x.ndim == 4 else x.unsqueeze(0)))
    given = [x for x in xs if x is not None]
    if not given and out is None:
        raise RuntimeError('grid_frames_uint8: nothing to write')
    if given:
        B = max(x.shape[0] for x in given) if out is None else out.shape[0]
        _, C, H, W = given[0].shape
    else:
        return out
    for x in given:
        if x.shape[1:] != (3, H, W) or x.shape[0] not in (1, B):
            raise RuntimeError('grid panels must be [B or 1,3,%d,%d], got %s' % (H, W, tuple(x.shape)))
    K = len(xs)
    if any(x is None for x in xs) and out is None:
        raise RuntimeError('grid_frames_uint8: a skipped panel needs the `out` tensor that already holds it')
    ptrs = (ctypes.c_void_p * K)(*[None if x is None else x.data_ptr() for x in xs])
    strides = (ctypes.c_int64 * K)(*[0 if (x is None or (x.shape[0] == 1 and B > 1)) else 3 * H * W for x in xs])
    if out is None:
        out = torch.empty(B, H, K * W, 3, device=given[0].device, dtype=torch.uint8)
    elif tuple(out.shape) != (B, H, K * W, 3) or out.dtype != torch.uint8 or not out.is_cuda or not out.is_contiguous():
        raise RuntimeError('grid_frames_uint8: out must be a contiguous uint8 [%d,%d,%d,3] device tensor' % (B, H, K * W))
    N.call('sgdfr_grid_to_u8_f32', ptrs, strides, K, N.ptr(out), B, H, W, int(bool(swap_rb)), N.stream())
    return out


def save_latent_codes(directory, names, latents):
    """The reference's W+ latent store: one `<name>.npy` holding a [n_latent,512] fp32 array per frame
    (invert_images.py:119-125).  `latents` [B,n_latent,512] comes back to the host in ONE copy."""
    lat = latents.detach().to('cpu', torch.float32).numpy()
    if lat.ndim != 3 or len(names) != lat.shape[0]:
        raise RuntimeError('save_latent_codes: %d names for latents of shape %s' % (len(names), tuple(lat.shape)))
    os.makedirs(directory, exist_ok=True)
    paths = []
    for name, code in zip(names, lat):
        path = os.path.join(directory, os.path.splitext(name)[0] + '.npy')
        np.save(path, code)
        paths.append(path)
    return paths


def load_latent_codes(paths, device=None):
    """Read per-frame `.npy` codes back into one [B,n_latent,512] tensor (pinned staging, one H2D copy)."""
    codes = [np.load(p) for p in paths]
    for p, c in zip(paths, codes):
        if c.ndim != 2 or c.shape != codes[0].shape or c.dtype != np.float32:
            raise RuntimeError('latent store %s: expected fp32 %s, got %s %s' % (p, codes[0].shape, c.dtype, c.shape))
    host = torch.from_numpy(np.stack(codes, 0))
    if device is None or torch.device(device).type == 'cpu':
        return host
    return host.pin_memory().to(device, non_blocking=True)


class ReenactmentSession:
    """One source identity, many target poses/expressions."""

    def __init__(self, G, A, source_code, truncation=0.7, trunc=None, batch=32, graph=False, shifts=None, streams=2):
        """shifts: a `shift.ShiftVectors` (direction tables of the dataset) -- then `render_targets` / `frames_for_targets`
        take the 3DMM parameters of source and targets and build the shift vectors on the device too.
        graph=True captures one `batch`-sized step (DirectionMatrix -> shift -> generator) in a hipGraph the first time a
        full batch is rendered and replays it afterwards: the ~65 launches of a step become one submission, which is what a
        small batch is bound by (B=1: 1.10 -> 0.72 ms per frame).  Weights must not be replaced while the graph is alive
        (call `reset_graph()` after loading new ones); partial last batches run eagerly.
        streams: consecutive chunks alternate between this many HIP streams (functional.StreamPipeline: the latency-bound head
        of chunk i+1 runs beside the big layers of chunk i; 1 = everything on the caller's stream).  Replayed graphs share
        their static buffers and stay on the caller's stream."""
        if source_code.ndim == 2:
            source_code = source_code.unsqueeze(0)
        if source_code.shape[0] != 1 or source_code.shape[1] != G.n_latent:
            raise RuntimeError('source_code must be one W+ code [1,%d,512]' % G.n_latent)
        if truncation < 1 and trunc is None:
            raise RuntimeError('truncation < 1 needs the truncation latent')
        self.G, self.A = G, A
        self.source = source_code.contiguous()
        self.truncation, self.trunc, self.batch = truncation, trunc, batch
        self.use_graph = bool(graph)
        self.shifts = shifts
        self._graph = None          # (hipGraph, static shift-vector input, static image output)
        self.n_streams = max(1, int(streams))
        self.pipeline_min_work = G.GRAPH_MAX_WORK      # batch * (size/256)^2 above which chunks alternate between the streams
        self._pipe = None

    def reset_graph(self):
        self._graph = None

    def _step(self, sv, image_out=None, no_graph=False, prefer_graph=False, verify=False):
        shift = self.A(sv)                                              # [b, L, 512] (w_plus) or [b, 512]
        b = sv.shape[0]
        w = self.source.expand(b, -1, -1).contiguous()
        layers = shift.shape[1] if shift.ndim == 3 else self.A.num_layers
        latent = F_.latent_prepare(w, self.G.n_latent, shift=shift, shift_layers=layers)
        # (a session with graph=True captures the whole step itself -- DirectionMatrix and latent shift included -- so the
        # generator's own per-forward graphs stay out of it)
        # (verify_range=False: the session checks every chunk's RangeToken itself, one chunk behind the launches; verify: the
        #  second rendering of a chunk that clamped -- the generator measures it, widens its range plan or falls back to bf16x3)
        img, _ = self.G([latent], input_is_latent=True, truncation=self.truncation, truncation_latent=self.trunc,
                        image_out=image_out, graph=False if (self.use_graph or no_graph or verify) else (True if prefer_graph else None),
                        verify_range=bool(verify))
        return img

    def _graphed_step(self, sv):
        mode = self.G.range_mode()
        if self._graph is not None and self._graph[3] != mode:          # the generator changed arithmetic (fallback / new weights)
            self._graph = None
        if self._graph is None:
            static_sv = sv.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                   # warm-up off the capture: packs, caches, allocator
                for _ in range(2):
                    self._step(static_sv)
            torch.cuda.current_stream().wait_stream(side)
            mode = self.G.range_mode()                                      # (the warm-up may have calibrated / fallen back)
            g = torch.cuda.CUDAGraph()
            with F_.capture_graph(g):
                static_out = self._step(static_sv)
            self._graph = (g, static_sv, static_out, mode)
        g, static_sv, static_out, mode = self._graph
        static_sv.copy_(sv)
        g.replay()
        # the captured launches add to the generator's saturation word like eager ones: snapshot it behind the replay
        tok = self.G._snapshot() if mode == 'fp16x3' else None
        return static_out.clone(), tok

    def _chunks(self, pipelined, device):
        """launch(lo, sv, u8) -> item / settle(item) -> (lo, sv, u8, images, graphed) for one chunk each; see _pipeline."""
        sess = self
        pipe = None
        # (chunks small enough for the generator's own hipGraph replay are host-bound and share the graph's static buffers: one stream)
        if pipelined and self.n_streams > 1 and not self.use_graph and \
                self.batch * (self.G.size / 256.0) ** 2 > self.pipeline_min_work:
            if self._pipe is None or self._pipe.streams[0].device != device:
                self._pipe = F_.StreamPipeline(self.n_streams, device)
            pipe = self._pipe

        class Chunks:
            def launch(self, lo, sv, u8):
                if sess.use_graph and sv.shape[0] == sess.batch:
                    img, tok = sess._graphed_step(sv)
                    return [lo, sv, u8, img, tok, True, None, 'graph']
                mode = sess.G.range_mode()
                if pipe is None:
                    # (one stream: the chunk's token is awaited right behind its launches, which exposes the ~0.5 ms of host enqueue
                    # time of an eager forward -- the generator replays its hipGraph for these, as a verified forward does)
                    img = sess._step(sv, u8, prefer_graph=True)
                    return [lo, sv, u8, img, sess.G.take_range_token(), False, None, mode]
                with pipe.next():
                    img = sess._step(sv, u8, no_graph=True)
                    tok = sess.G.take_range_token()
                return [lo, sv, u8, img, tok, False, pipe.last, mode]

            def settle(self, item):
                lo, sv, u8, img, tok, graphed, stream, mode = item
                # False for a chunk that clamped operands AND for one that was in flight beside it on the other stream (the shared
                # saturation word cannot tell the two apart: RangeToken.suspect)
                ok = sess.G.range_ok(tok)
                if stream is not None:
                    pipe.join(img, stream=stream)                           # the caller's stream may now read this chunk
                if not ok:
                    sess._graph = None                                      # this chunk again, eagerly and VERIFIED: the generator measures
                    img, graphed = sess._step(sv, u8, verify=True), False   # it and widens its range plan (or renders in bf16x3)
                return lo, sv, u8, img, graphed

        return Chunks()

    def _pipeline(self, shift_vectors, target_for=None):
        """(lo, image batch) for every chunk of `shift_vectors`, each VERIFIED against the generator's fp16 range plan before
        it is yielded: chunk i+1 is queued first (on the other HIP stream when the session pipelines), then chunk i's RangeToken
        is awaited (the GPU never idles for the check), and a chunk that clamped operands is rendered again in the generator's
        fallback arithmetic.  target_for(lo, b) -> functional.U8Target or None (eager chunks: the last ToRGB launch writes uint8
        frames)."""
        n = shift_vectors.shape[0]
        chunks = self._chunks(n > self.batch, shift_vectors.device)
        pending = None
        for lo in range(0, n, self.batch):
            sv = shift_vectors[lo:lo + self.batch]
            cur = chunks.launch(lo, sv, target_for(lo, sv.shape[0]) if target_for is not None else None)
            if pending is not None:
                yield chunks.settle(pending)
            pending = cur
        if pending is not None:
            yield chunks.settle(pending)

    def streaming(self, as_uint8=False):
        """For frames that arrive chunk by chunk (a live video, bench.py's inference config): `push(shift_vectors [b, dim])`
        queues that chunk and returns the images of the PREVIOUS one (None for the first), `flush()` returns the last -- the same
        one-chunk look-ahead, stream alternation and range verification as `frames`, kept alive between calls."""
        sess = self

        class Streaming:
            def __init__(self):
                self.chunks, self.pending, self.count = None, None, 0

            def _out(self, item):
                lo, sv, u8, img, graphed = self.chunks.settle(item)
                return images_to_uint8(img) if (as_uint8 and graphed) else img

            @torch.no_grad()
            def push(self, shift_vectors):
                if self.chunks is None:
                    self.chunks = sess._chunks(True, shift_vectors.device)
                cur = self.chunks.launch(self.count, shift_vectors, F_.U8Target() if as_uint8 else None)
                self.count += shift_vectors.shape[0]
                prev, self.pending = self.pending, cur
                return self._out(prev) if prev is not None else None

            @torch.no_grad()
            def flush(self):
                prev, self.pending = self.pending, None
                return self._out(prev) if prev is not None else None

        return Streaming()

    @torch.no_grad()
    def frames(self, shift_vectors, as_uint8=False):
        """shift_vectors [N, input_dim] -> generator of image batches (same result as N generate_image calls)."""
        for lo, sv, u8, img, graphed in self._pipeline(shift_vectors, (lambda lo, b: F_.U8Target()) if as_uint8 else None):
            # eager uint8 frames come straight out of the last ToRGB launch (no fp32 image is stored); a replayed graph
            # produced fp32 images
            yield images_to_uint8(img) if (as_uint8 and graphed) else img

    def render(self, shift_vectors, as_uint8=False):
        return torch.cat(list(self.frames(shift_vectors, as_uint8=as_uint8)), 0)

    def shift_vectors_for(self, angles_source, params_source, angles_target, params_target):
        """[N, learned_directions] for N target frames against the one source identity, in one launch and without a
        host round trip (the reference: run_inference.py:178 `make_shift`, once per frame, ~10 `.cpu()` syncs each)."""
        if self.shifts is None:
            raise RuntimeError('this session was built without direction tables (pass shifts=ShiftVectors(...))')
        return self.shifts.make_shift(angles_source, angles_target, params_source, params_target)

    def frames_for_targets(self, angles_source, params_source, angles_target, params_target, as_uint8=False):
        """run_inference.py:170-181 for all target frames: 3DMM parameters in, image batches out."""
        return self.frames(self.shift_vectors_for(angles_source, params_source, angles_target, params_target), as_uint8=as_uint8)

    def render_targets(self, angles_source, params_source, angles_target, params_target, as_uint8=False):
        return torch.cat(list(self.frames_for_targets(angles_source, params_source, angles_target, params_target,
                                                      as_uint8=as_uint8)), 0)

    @torch.no_grad()
    def video_frames(self, source_image, target_images, shift_vectors, swap_rb=True):
        """source | target | reenacted frames [N,H,3W,3] uint8 (the --save_video output of run_inference.py:186-198)."""
        n = shift_vectors.shape[0]
        H, W = source_image.shape[-2], source_image.shape[-1]
        video = torch.empty(n, H, 3 * W, 3, device=shift_vectors.device, dtype=torch.uint8)
        # eager chunks: the reenacted panel is written by the generator's last launch; the other two by one grid launch
        target_for = lambda lo, b: F_.U8Target(video[lo:lo + b], panel=2, swap_rb=swap_rb)
        for lo, sv, u8, img, graphed in self._pipeline(shift_vectors, target_for):
            frames = video[lo:lo + sv.shape[0]]
            grid_frames_uint8([source_image, target_images[lo:lo + sv.shape[0]], img if graphed else None], swap_rb=swap_rb, out=frames)
        return video
