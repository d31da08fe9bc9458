"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the reference's shift-vector
construction, in the reference's own formulation -- per-frame numpy scalars for inference, float32 torch tensor ops for
the trainer.  Pinned by tests/golden/kat8_shift.npz, which oracle/make_golden_shift.py generates by running the REAL
reference functions (imported from /root/reference in the build container) and asserting bit-equality with this file.

  initialize_directions   libs/utilities/generic.py:36-114
  make_shift              run_inference.py:201-254            (Inference.make_shift)
  make_shift_vector       libs/utilities/utils_train.py:127-175
  make_shift_vector_50    libs/utilities/utils_train.py:177-288
"""
import numpy as np
import torch

# libs/configs/config_directions.py:5-27
_CFG = {'voxceleb': (0, 1, 2, (40, 20, 20)), 'ffhq': (0, 1, -1, (40, 20, 20))}


def initialize_directions(dataset_type, learned_directions, shift_scale, ranges):
    """generic.py:36-114 with the ranges array passed in (the reference np.load()s it, config_directions.py:29-39)."""
    yaw, pitch, roll, scales = _CFG['voxceleb' if dataset_type == 'voxceleb' else 'ffhq']
    ranges = np.asarray(ranges).astype('float64')
    jaw_range = ranges[3]
    exp_ranges = ranges[4:]
    angle_scales = np.zeros(3)
    angle_scales[:] = scales
    angle_directions = np.zeros(3)
    angle_directions[:] = (yaw, pitch, roll)
    count_pose = sum(1 for d in angle_directions if d != -1) + 1
    num_expressions = learned_directions - count_pose
    directions_exp = []
    for i in range(num_expressions):
        x = (exp_ranges[i][0], exp_ranges[i][1])
        y = (-shift_scale, shift_scale)
        A = np.vstack([x, np.ones(len(x))]).T
        m, c = np.linalg.lstsq(A, y, rcond=None)[0]                      # generic.py:91-96
        directions_exp.append({'exp_component': i, 'A_direction': i + count_pose, 'a': m, 'b': c})
    A = np.vstack([(jaw_range[0], jaw_range[1]), np.ones(2)]).T
    a_jaw, b_jaw = np.linalg.lstsq(A, (-6, 6), rcond=None)[0]            # generic.py:100-105
    return dict(count_pose=count_pose, num_expressions=num_expressions, directions_exp=directions_exp, a_jaw=a_jaw,
                b_jaw=b_jaw, angle_scales=angle_scales, yaw=yaw, pitch=pitch, roll=roll, shift_scale=shift_scale,
                learned_directions=learned_directions)


def make_shift(cfg, angles_source, angles_target, params_source, params_target):
    """run_inference.py:201-254 for ONE frame: inputs are float32 torch tensors [1,3] / {'pose': [1,6], 'alpha_exp': [1,50]};
    returns [1, learned_directions] float32.  numpy scalar arithmetic as there (float32 0-d values against float64
    scalars), one rounding when the value is stored into the float32 tensor."""
    out = torch.zeros(cfg['learned_directions'])
    sc = cfg['shift_scale']
    for k in range(3):                                                   # :217-236 (rows 0, 1, 2)
        s = angles_source[:, k][0].numpy() * sc / cfg['angle_scales'][k]
        t = angles_target[:, k][0].numpy() * sc / cfg['angle_scales'][k]
        out[k] = t - s
    a, b = cfg['a_jaw'], cfg['b_jaw']                                   # :237-245 (row 3)
    out[3] = (a * params_target['pose'][0, 3].numpy() + b) - (a * params_source['pose'][0, 3].numpy() + b)
    es, et = params_source['alpha_exp'][0].numpy(), params_target['alpha_exp'][0].numpy()
    for index in range(cfg['num_expressions']):                          # :246-254
        d = cfg['directions_exp'][index]
        out[index + cfg['count_pose']] = (d['a'] * et[d['exp_component']] + d['b']) - (d['a'] * es[d['exp_component']] + d['b'])
    return out.unsqueeze(0)


def _rows_full(cfg, out, rows, param_source, param_target, angles_source, angles_target):
    sc = cfg['shift_scale']
    for col, d in enumerate((cfg['yaw'], cfg['pitch'], cfg['roll'])):   # utils_train.py:132-148 / :185-201
        if d != -1:
            out[rows, d] = angles_target[rows, col] * sc / cfg['angle_scales'][col] - \
                angles_source[rows, col] * sc / cfg['angle_scales'][col]
    a, b = cfg['a_jaw'], cfg['b_jaw']                                   # :150-157 / :203-210
    out[rows, cfg['count_pose'] - 1] = (a * param_target['pose'][rows, 3] + b) - (a * param_source['pose'][rows, 3] + b)
    for index in range(cfg['num_expressions']):                          # :159-172 / :212-225
        d = cfg['directions_exp'][index]
        c = d['exp_component']
        out[rows, index + cfg['count_pose']] = (d['a'] * param_target['alpha_exp'][rows, c] + d['b']) - \
            (d['a'] * param_source['alpha_exp'][rows, c] + d['b'])


def make_shift_vector(cfg, param_source, param_target, angles_source, angles_target):
    """utils_train.py:127-175: float32 torch tensors, one op per direction."""
    B = angles_source.shape[0]
    out = torch.zeros(B, cfg['learned_directions'])
    _rows_full(cfg, out, slice(0, B), param_source, param_target, angles_source, angles_target)
    return out


def make_shift_vector_50(cfg, param_source, param_target, angles_source, angles_target, target_indices, u):
    """utils_train.py:177-288 with the random draws injected: target_indices [B/2] ints (np.random.choice there),
    u [B/2] float32 in [0,1) (one torch.rand(1) per sample there)."""
    B = angles_source.shape[0]
    assert B % 2 == 0
    h = B // 2
    out = torch.zeros(B, cfg['learned_directions'])
    _rows_full(cfg, out, slice(0, h), param_source, param_target, angles_source, angles_target)
    sc = cfg['shift_scale']
    for count, batch in enumerate(range(h, B)):                          # :231-286
        ind = int(target_indices[count])
        start = None
        for col, d in enumerate((cfg['yaw'], cfg['pitch'], cfg['roll'])):
            if start is None and ind == d:
                start = angles_source[batch, col] * sc / cfg['angle_scales'][col]
        if start is None and ind == cfg['count_pose'] - 1:
            start = cfg['a_jaw'] * param_source['pose'][batch, 3] + cfg['b_jaw']
        if start is None:
            d = next((d for d in cfg['directions_exp'] if d['A_direction'] == ind), None)
            if d is None:
                continue
            start = d['a'] * param_source['alpha_exp'][batch][d['exp_component']] + d['b']
        lo, hi = (-sc - start), (sc - start)
        out[batch, ind] = (lo - hi) * u[count] + hi
    return out
