"""Generator fine-tuning on one source (PTI), the counterpart of libs/optimization.py:25-72 (`optimize_g`): same parameter
selection, optimiser and truncation; the perceptual loss is the caller's (LPIPS and the losses around it are neighbours of
the hot path).  The step -- forward, backward through autograd.SynthesisFn, Adam -- can be captured once and
replayed as a hipGraph: at one source per step the eager step is bound by its host launches (~5 ms), the replay takes
3.4-3.8 ms (bench.py --config pti, scripts/pti_step_bench.py)."""
import torch

from . import _native as N
from . import functional as F_


def pti_parameters(generator, optimize_all=False):
    """optimization.py:31-40: convs[4..11] (pt_l2_lambda 100) or every parameter (pt_l2_lambda 1)."""
    if optimize_all:
        return list(generator.parameters()), 1
    return [p for i in range(11, 3, -1) for p in generator.convs[i].parameters()], 100


def l2_loss_fn(imgs_gen, real_imgs, pt_l2_lambda):
    """The L2 term of calc_loss (libs/criteria/l2_loss.py: mse) weighted as optimization.py does; stand-in default."""
    return pt_l2_lambda * torch.nn.functional.mse_loss(imgs_gen, real_imgs)


class FusedAdam:
    """torch.optim.Adam(params, lr) as libs/optimization.py:41 builds it (default betas and eps, no weight decay) with the whole
    update in ONE launch (sgdfr_adam_f32) and the step count on the device: under a hipGraph torch's capturable multi-tensor Adam
    spends ~100 launches on the 24 tensors of a PTI step (per-parameter step-size tensors), a tenth of the step.  Interface: the
    three calls the step uses -- zero_grad(set_to_none), step(), and state in .state for inspection."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params or len(self.params) > N.MAX_ADAM_TENSORS:
            raise ValueError('FusedAdam takes 1..%d parameter tensors' % N.MAX_ADAM_TENSORS)
        for p in self.params:
            N.require_device(p)
            if not p.is_contiguous():
                raise ValueError('FusedAdam: parameters must be contiguous')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = self.params[0].device
        self.step_count = torch.zeros(1, device=dev, dtype=torch.float32)
        self.state = {p: {'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)} for p in self.params}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        self.step_count.add_(1.0)
        arr = (N.AdamTensor * len(live))()
        keep = []
        for i, p in enumerate(live):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            N.require_device(g)
            st = self.state[p]
            keep.append(g)
            arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n = p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()
        N.call('sgdfr_adam_f32', arr, len(live), N.ptr(self.step_count), self.lr, self.betas[0], self.betas[1], self.eps, N.stream())
        for p in live:          # the native launch wrote p behind autograd's back: bump the version counter the weight packs
            torch.autograd.graph.increment_version(p)       # (ModulatedConv2d.packed) and the saved-tensor checks watch -- no launch


class GraphedStep:
    """Captures `step_fn()` (a full training step on static tensors: forward, backward, optimizer.step of an optimiser built
    with capturable=True) after `warmup` eager calls on a side stream, then replays it.  Call it like step_fn; the returned
    tensor is the static output of the captured step."""

    def __init__(self, step_fn, warmup=3, clear_grads_of=()):
        """clear_grads_of: parameters whose .grad is set to None right before the capture, so that the captured backward WRITES their
        gradients (into buffers of the graph's pool) instead of adding to tensors left by the warm-up steps -- one in-place add per
        parameter tensor and step otherwise (the whole-network capture recipe of torch.cuda.graphs)."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                step_fn()
        torch.cuda.current_stream().wait_stream(side)
        for p in clear_grads_of:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with F_.capture_graph(self.graph):
            self.out = step_fn()          # recorded, not executed: the first replay is step warmup + 1
        self.steps_done = warmup

    def __call__(self):
        self.graph.replay()
        self.steps_done += 1
        return self.out


def optimize_g(generator, latent, real_imgs, trunc, opt_steps=200, lr=3e-3, optimize_all=False, loss_fn=None, truncation=0.7,
               graph=True, freeze_unused=False, fused_adam=True):
    """Fine-tunes `generator` in place so that G(latent) reproduces real_imgs (optimization.py:25-72).  `trunc` is the
    truncation latent (the reference draws generator.mean_latent(4096) itself); loss_fn(imgs_gen, real_imgs, pt_l2_lambda)
    -> scalar (default: the weighted L2 term).  freeze_unused=True stops producing gradients nobody reads (the reference
    leaves requires_grad on every parameter).  fused_adam: the one-launch Adam (FusedAdam; False = torch.optim.Adam as the reference
    builds it -- the same update to rounding).  Returns (generator, last loss tensor)."""
    params, pt_l2_lambda = pti_parameters(generator, optimize_all)
    loss_fn = loss_fn or l2_loss_fn
    saved = None
    if freeze_unused:
        ids = {id(p) for p in params}
        saved = [(p, p.requires_grad) for p in generator.parameters()]
        for p in generator.parameters():
            p.requires_grad_(id(p) in ids)
    generator.train()
    if fused_adam and len(params) <= N.MAX_ADAM_TENSORS and all(p.is_cuda and p.is_contiguous() for p in params):
        optimizer = FusedAdam(params, lr=lr)
    else:
        optimizer = torch.optim.Adam(params, lr=lr, capturable=bool(graph))
    latent, real_imgs = latent.detach(), real_imgs.detach()

    def step():
        imgs_gen, _ = generator([latent], input_is_latent=True, return_latents=False, truncation=truncation,
                                truncation_latent=trunc)
        loss = loss_fn(imgs_gen, real_imgs, pt_l2_lambda)
        optimizer.zero_grad(set_to_none=True)         # (optimization.py:66: the backward then writes fresh gradients, no zero + add pair)
        loss.backward()
        optimizer.step()
        return loss.detach()

    loss = None
    try:
        if graph and opt_steps > 4:
            runner = GraphedStep(step, warmup=3, clear_grads_of=list(generator.parameters()))
            for _ in range(opt_steps - runner.steps_done):
                loss = runner()
        else:
            for _ in range(opt_steps):
                loss = step()
    finally:
        if saved is not None:
            for p, rg in saved:
                p.requires_grad_(rg)
    return generator, loss
