"""Generator fine-tuning on one source (PTI), the counterpart of libs/optimization.py:25-72 (`optimize_g`): same parameter
selection, optimiser and truncation; the perceptual loss is the caller's (LPIPS and the losses around it are neighbours of
the hot path).  The step -- forward, backward through autograd.SynthesisFn, Adam -- can be captured once and
replayed as a hipGraph: at one source per step the eager step is bound by its host launches (~5 ms), the replay takes
3.4-3.8 ms (bench.py --config pti, scripts/pti_step_bench.py)."""
import torch

from . import functional as F_


def pti_parameters(generator, optimize_all=False):
    """optimization.py:31-40: convs[4..11] (pt_l2_lambda 100) or every parameter (pt_l2_lambda 1)."""
    if optimize_all:
        return list(generator.parameters()), 1
    return [p for i in range(11, 3, -1) for p in generator.convs[i].parameters()], 100


def l2_loss_fn(imgs_gen, real_imgs, pt_l2_lambda):
    """The L2 term of calc_loss (libs/criteria/l2_loss.py: mse) weighted as optimization.py does; stand-in default."""
    return pt_l2_lambda * torch.nn.functional.mse_loss(imgs_gen, real_imgs)


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
               graph=True, freeze_unused=False):
    """Fine-tunes `generator` in place so that G(latent) reproduces real_imgs (optimization.py:25-72).  `trunc` is the
    truncation latent (the reference draws generator.mean_latent(4096) itself); loss_fn(imgs_gen, real_imgs, pt_l2_lambda)
    -> scalar (default: the weighted L2 term).  freeze_unused=True stops producing gradients nobody reads (the reference
    leaves requires_grad on every parameter).  Returns (generator, last loss tensor)."""
    params, pt_l2_lambda = pti_parameters(generator, optimize_all)
    loss_fn = loss_fn or l2_loss_fn
    saved = None
    if freeze_unused:
        ids = {id(p) for p in params}
        saved = [(p, p.requires_grad) for p in generator.parameters()]
        for p in generator.parameters():
            p.requires_grad_(id(p) in ids)
    generator.train()
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
