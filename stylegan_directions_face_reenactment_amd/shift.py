"""Shift vectors (the DirectionMatrix input) built on the GPU for whole batches -- SURVEY.md §8f-2.

Counterparts, same names and argument order:
  * ``initialize_directions``            libs/utilities/generic.py:36-114
  * ``ShiftVectors.make_shift``          run_inference.py:201-254   (Inference.make_shift; per frame, ~10 host syncs there)
  * ``ShiftVectors.make_shift_vector``   libs/utilities/utils_train.py:127-175
  * ``ShiftVectors.make_shift_vector_50``libs/utilities/utils_train.py:177-288

The reference pulls every 3DMM parameter to the host (``.detach().cpu().numpy()``), does scalar arithmetic there and
uploads the 15 numbers again -- once per target frame.  Here ``params_*`` / ``angles_*`` stay device tensors for the whole
batch ([N,3] angles, ``params['pose']`` [N,6], ``params['alpha_exp']`` [N,50]) and one launch of ``sgdfr_make_shift_f32``
writes the [N, learned_directions] matrix; nothing synchronises.  The per-direction recipe (which parameter, which affine
map) is a small host table handed to the kernel by value.  Both reference call sites' arithmetic is mirrored operation by
operation (float64 numpy scalars in run_inference.py, float32 tensors in utils_train.py; csrc/shift.hip) and is bit-identical
to goldens captured from the real functions run on CPU tensors (tests/test_gpu_shift.py); against the reference on a CUDA
device the float32 path can differ by 1 ulp where that device's division is not correctly rounded.

Inputs: device tensors (float32), or numpy arrays / lists, which are uploaded to the current device; CPU torch tensors are
refused like everywhere in this package.  One deliberate difference in return types: `make_shift_vector_50` returns
`target_indices` as an int32 DEVICE tensor (no host sync in the training step) where utils_train.py:184-188 returns a host
numpy array -- pass `indices_on_host=True` for the reference's type.
"""
import os

import numpy as np
import torch

from . import _native as N

# libs/configs/config_directions.py:5-27 (which direction of A each pose angle drives, and the degrees that map to shift_scale)
DATASET_DIRECTIONS = {
    'voxceleb': dict(yaw_direction=0, pitch_direction=1, roll_direction=2, jaw_direction=3, yaw_scale=40, pitch_scale=20,
                     roll_scale=20, ranges_filepath='./libs/configs/ranges_voxceleb.npy'),
    'ffhq': dict(yaw_direction=0, pitch_direction=1, roll_direction=-1, jaw_direction=3, yaw_scale=40, pitch_scale=20,
                 roll_scale=20, ranges_filepath='./libs/configs/ranges_FFHQ.npy'),
}


def get_direction_ranges(range_filepath):
    """[54,2] float64 (min, max) of yaw, pitch, roll, jaw and the 50 expression coefficients (config_directions.py:29-39);
    raises instead of the reference's print + exit()."""
    if not os.path.exists(range_filepath):
        raise FileNotFoundError('{} does not exist'.format(range_filepath))
    return np.asarray(np.load(range_filepath)).astype('float64')


def _line(x0, y0, x1, y1):
    """(slope, intercept) through two points, solved the way generic.py:91-95 does (least squares on the 2x2 system) so
    the float64 coefficients carry the same rounding."""
    A = np.vstack([(x0, x1), np.ones(2)]).T
    m, c = np.linalg.lstsq(A, (y0, y1), rcond=None)[0]
    return m, c


def initialize_directions(dataset_type, learned_directions, shift_scale, ranges=None):
    """(count_pose, num_expressions, directions_exp, jaw_dict, angle_scales, angle_directions) as generic.py:36-114.
    `ranges`: the [54,2] array (or a path to the .npy); default = the reference's relative path for the dataset."""
    cfg = DATASET_DIRECTIONS['voxceleb' if dataset_type == 'voxceleb' else 'ffhq']
    if ranges is None:
        ranges = cfg['ranges_filepath']
    if isinstance(ranges, (str, os.PathLike)):
        ranges = get_direction_ranges(ranges)
    ranges = np.asarray(ranges, dtype='float64')
    min_jaw, max_jaw = ranges[3][0], ranges[3][1]
    exp_ranges = ranges[4:]
    angle_scales = np.array([cfg['yaw_scale'], cfg['pitch_scale'], cfg['roll_scale']], dtype=np.float64)
    angle_directions = np.array([cfg['yaw_direction'], cfg['pitch_direction'], cfg['roll_direction']], dtype=np.float64)
    count_pose = int(sum(1 for d in angle_directions if d != -1)) + 1          # + jaw
    num_expressions = learned_directions - count_pose
    directions_exp = []
    for i in range(num_expressions):
        lo, hi = exp_ranges[i][0], exp_ranges[i][1]
        a, b = _line(lo, -shift_scale, hi, shift_scale)
        directions_exp.append({'exp_component': i, 'A_direction': i + count_pose, 'max_shift': hi, 'min_shift': lo,
                               'a': a, 'b': b})
    a_jaw, b_jaw = _line(min_jaw, -6, max_jaw, 6)               # generic.py:100: fixed +-6, not shift_scale
    jaw_dict = {'a': a_jaw, 'b': b_jaw, 'max': max_jaw, 'min': min_jaw}
    return count_pose, num_expressions, directions_exp, jaw_dict, angle_scales, angle_directions


def _table(entries, D):
    if D > N.MAX_DIRECTIONS:
        raise RuntimeError('at most %d learned directions, got %d' % (N.MAX_DIRECTIONS, D))
    arr = (N.Direction * D)()
    for k in range(D):
        arr[k].kind, arr[k].col, arr[k].a, arr[k].b = N.DIR_ZERO, 0, 0.0, 0.0
    for k, kind, col, a, b in entries:          # later entries overwrite earlier ones, like the reference's assignments
        if 0 <= k < D:
            arr[k].kind, arr[k].col, arr[k].a, arr[k].b = kind, int(col), float(a), float(b)
    return arr


def _dev(t, like=None):
    if not isinstance(t, torch.Tensor):     # numpy / lists: uploaded next to `like`, or to the current device
        t = torch.as_tensor(np.asarray(t), dtype=torch.float32)
        t = t.to(like.device if like is not None else torch.device('cuda', torch.cuda.current_device()))
    N.require_device(t)
    return N.f32c(t)


class ShiftVectors:
    """Holds the direction tables of one (dataset, learned_directions, shift_scale) setting and builds shift vectors on
    the GPU.  Attribute names follow the reference objects (Inference / Utilities_train): count_pose, num_expressions,
    directions_exp, a_jaw, b_jaw, max_jaw, min_jaw, angle_scales, angle_directions, shift_scale, learned_directions."""

    def __init__(self, dataset_type='voxceleb', learned_directions=15, shift_scale=6, ranges=None):
        self.dataset_type, self.learned_directions, self.shift_scale = dataset_type, int(learned_directions), shift_scale
        (self.count_pose, self.num_expressions, self.directions_exp, jaw, self.angle_scales,
         self.angle_directions) = initialize_directions(dataset_type, learned_directions, shift_scale, ranges)
        self.a_jaw, self.b_jaw, self.max_jaw, self.min_jaw = jaw['a'], jaw['b'], jaw['max'], jaw['min']
        cfg = DATASET_DIRECTIONS['voxceleb' if dataset_type == 'voxceleb' else 'ffhq']
        self.yaw_direction, self.pitch_direction, self.roll_direction = (cfg['yaw_direction'], cfg['pitch_direction'],
                                                                          cfg['roll_direction'])
        D, sc = self.learned_directions, self.shift_scale
        exps = [(e['A_direction'], N.DIR_EXP, e['exp_component'], e['a'], e['b']) for e in self.directions_exp]
        # run_inference.py:217-252 writes yaw, pitch, roll to rows 0, 1, 2 and the jaw to row 3 whatever the dataset, then the
        # expressions from row count_pose on
        self._table_inference = _table(
            [(0, N.DIR_ANGLE, 0, sc, self.angle_scales[0]), (1, N.DIR_ANGLE, 1, sc, self.angle_scales[1]),
             (2, N.DIR_ANGLE, 2, sc, self.angle_scales[2]), (3, N.DIR_JAW, 3, self.a_jaw, self.b_jaw)] + exps, D)
        # utils_train.py:132-172 honours the per-dataset direction rows (-1 = absent) and puts the jaw at count_pose - 1
        pose = [(d, N.DIR_ANGLE, c, sc, self.angle_scales[c])
                for c, d in enumerate((self.yaw_direction, self.pitch_direction, self.roll_direction)) if d != -1]
        self._table_train = _table(pose + [(self.count_pose - 1, N.DIR_JAW, 3, self.a_jaw, self.b_jaw)] + exps, D)

    # ------------------------------------------------------------------ launches
    def _launch(self, table, arith, angles_source, angles_target, params_source, params_target, out=None):
        at = _dev(angles_target)
        n = at.shape[0]
        pt, et = _dev(params_target['pose'], at), _dev(params_target['alpha_exp'], at)
        as_, ps, es = _dev(angles_source, at), _dev(params_source['pose'], at), _dev(params_source['alpha_exp'], at)
        if at.ndim != 2 or at.shape[1] < 3 or pt.shape[0] != n or et.shape[0] != n:
            raise RuntimeError('make_shift: target angles %s / pose %s / alpha_exp %s do not describe %d frames'
                               % (tuple(at.shape), tuple(pt.shape), tuple(et.shape), n))
        strides = []
        for name, s, t in (('angles', as_, at), ('pose', ps, pt), ('alpha_exp', es, et)):
            if s.ndim != 2 or s.shape[1] != t.shape[1] or s.shape[0] not in (1, n):
                raise RuntimeError('make_shift: source %s %s does not match target %s' % (name, tuple(s.shape), tuple(t.shape)))
            strides.append(0 if (s.shape[0] == 1 and n != 1) else s.shape[1])
        if at.shape[1] != 3:
            raise RuntimeError('make_shift: angles must be [N,3] (yaw, pitch, roll), got %s' % (tuple(at.shape),))
        if out is None:
            out = torch.empty(n, self.learned_directions, device=at.device, dtype=torch.float32)
        N.call('sgdfr_make_shift_f32', N.ptr(as_), strides[0], N.ptr(ps), strides[1], N.ptr(es), strides[2], N.ptr(at),
               N.ptr(pt), N.ptr(et), pt.shape[1], et.shape[1], table, self.learned_directions, N.ptr(out), n, arith, N.stream())
        return out

    def make_shift(self, angles_source, angles_target, params_source, params_target):
        """run_inference.py:201-254 for N target frames at once: [N, learned_directions].  The source may be one identity
        ([1,.] tensors, broadcast) or one per frame."""
        return self._launch(self._table_inference, 0, angles_source, angles_target, params_source, params_target)

    def make_shift_vector(self, param_source, param_target, angles_source, angles_target):
        """utils_train.py:127-175: [B, learned_directions] in float32 tensor arithmetic."""
        return self._launch(self._table_train, 1, angles_source, angles_target, param_source, param_target)

    def make_shift_vector_50(self, param_source, param_target, angles_source, angles_target, target_indices=None, u=None,
                             indices_on_host=False):
        """utils_train.py:177-288: first half of the batch = full reenactment shift, second half = one random direction
        each.  Returns (shift_vector [B,D], target_indices [B/2]): the indices as an int32 device tensor, or -- with
        indices_on_host=True -- as the host numpy array the reference returns (one device->host sync).  `target_indices` /
        `u` may be given (the reference draws them with np.random.choice / torch.rand); by default they are drawn on the
        device."""
        ang_s, ang_t = _dev(angles_source), _dev(angles_target)
        B = ang_s.shape[0]
        if B % 2 != 0:
            raise RuntimeError('Batch size should be even number!')          # utils_train.py:179-181 (print + exit there)
        h, D = B // 2, self.learned_directions
        ps, es = _dev(param_source['pose'], ang_s), _dev(param_source['alpha_exp'], ang_s)
        pt, et = _dev(param_target['pose'], ang_s), _dev(param_target['alpha_exp'], ang_s)
        out = torch.empty(B, D, device=ang_s.device, dtype=torch.float32)        # both launches write their half in place
        self._launch(self._table_train, 1, ang_s[:h], ang_t[:h], {'pose': ps[:h], 'alpha_exp': es[:h]},
                     {'pose': pt[:h], 'alpha_exp': et[:h]}, out=out[:h])
        if target_indices is None:
            target_indices = torch.randint(0, D, (h,), device=ang_s.device, dtype=torch.int32)
        elif not isinstance(target_indices, torch.Tensor):
            target_indices = torch.as_tensor(np.asarray(target_indices), dtype=torch.int32).to(ang_s.device)
        which = target_indices.to(device=ang_s.device, dtype=torch.int32).contiguous()
        u = torch.rand(h, device=ang_s.device, dtype=torch.float32) if u is None else _dev(u, ang_s).reshape(-1)
        if which.numel() != h or u.numel() != h:
            raise RuntimeError('make_shift_vector_50: need %d target_indices / draws' % h)
        a2, p2, e2 = ang_s[h:], ps[h:], es[h:]                                    # row slices of contiguous tensors
        N.call('sgdfr_make_shift_random_f32', N.ptr(a2), N.ptr(p2), N.ptr(e2), p2.shape[1], e2.shape[1],
               N.ptr(which), N.ptr(u), float(self.shift_scale), self._table_train, D, N.ptr(out[h:]), h, N.stream())
        return out, (which.cpu().numpy() if indices_on_host else which)
