// Shift-vector construction on the device: (3DMM parameters of source and target) -> rows of the DirectionMatrix input.
//
// The reference builds this vector on the host, one frame at a time, with ~10 `.detach().cpu().numpy()` round trips
// (run_inference.py:201-254 Inference.make_shift), and in the trainer with one small torch op per direction
// (libs/utilities/utils_train.py:127-175 make_shift_vector, :177-288 make_shift_vector_50).  Here a whole batch is one
// launch, one thread per (frame, direction); the per-direction recipe is a small table passed by value.
//
// The two reference call sites do NOT use the same arithmetic, and both are mirrored operation by operation so the
// result is bit-identical to the one the caller would have computed:
//   arith 0 (run_inference.py): 0-d float32 numpy values against float64 numpy scalars ->  the angle is scaled in
//           float32 (`yaw * shift_scale`), divided in float64 (`/ angle_scales[i]`), differences are taken in float64 and
//           rounded once when stored into the float32 tensor; `a*x + b` is float64 throughout.
//   arith 1 (utils_train.py): float32 torch tensors against Python/numpy scalars -> every operation rounds to float32
//           (scalar operands are cast to float32 first), multiply and add are separate roundings (no fma).
#include "common.h"

namespace sgdfr {

struct DirTable {
    sgdfr_direction d[SGDFR_MAX_DIRECTIONS];
};

// the three parameter arrays of one face set, indexed by (kind - 1): angles [*,3] (yaw, pitch, roll), pose [*,pose_dim],
// alpha_exp [*,exp_dim]; bs = batch stride in floats (0 = one source for every frame)
struct ShiftSrc {
    const float* base[3];
    int64_t bs[3];
};

// (An if-chain over three (pointer, stride) members was miscompiled by hipcc 7.2 for gfx950: the expression branch kept
// the pose stride -- caught by the golden test.  Both selections are therefore written as explicit, separate selects.)
__device__ __forceinline__ float pick(const ShiftSrc& p, const sgdfr_direction& e, int n) {
    const int k = e.kind;
    const float* b = k == SGDFR_DIR_ANGLE ? p.base[0] : (k == SGDFR_DIR_JAW ? p.base[1] : p.base[2]);
    const int64_t st = k == SGDFR_DIR_ANGLE ? p.bs[0] : (k == SGDFR_DIR_JAW ? p.bs[1] : p.bs[2]);
    return b[n * st + e.col];
}

// position of a value on the direction's shift axis, float64 flavour (run_inference.py:217-252)
__device__ __forceinline__ double place64(const sgdfr_direction& e, float x) {
    if (e.kind == SGDFR_DIR_ANGLE) return __ddiv_rn((double)__fmul_rn(x, (float)e.a), e.b);
    return __dadd_rn(__dmul_rn(e.a, (double)x), e.b);
}

// float32 flavour (utils_train.py:132-172)
__device__ __forceinline__ float place32(const sgdfr_direction& e, float x) {
    if (e.kind == SGDFR_DIR_ANGLE) return __fdiv_rn(__fmul_rn(x, (float)e.a), (float)e.b);
    return __fadd_rn(__fmul_rn((float)e.a, x), (float)e.b);
}

__global__ __launch_bounds__(256) void make_shift_kernel(ShiftSrc src, ShiftSrc tgt, DirTable tab, int D, float* __restrict__ out,
                                                        int N, int arith) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, k = i - n * D;
    const sgdfr_direction e = tab.d[k];
    float v = 0.f;
    if (e.kind != SGDFR_DIR_ZERO) {
        const float xs = pick(src, e, n), xt = pick(tgt, e, n);
        if (arith == 0) v = (float)__dsub_rn(place64(e, xt), place64(e, xs));
        else v = __fsub_rn(place32(e, xt), place32(e, xs));
    }
    out[i] = v;
}

// second half of make_shift_vector_50 (utils_train.py:227-286): one randomly chosen direction per sample gets a uniform
// draw from [-shift_scale - start, shift_scale - start]; `u` in [0,1) and the direction indices come from the caller
__global__ __launch_bounds__(256) void make_shift_random_kernel(ShiftSrc src, DirTable tab, int D, const int* __restrict__ which,
                                                               const float* __restrict__ u, float shift_scale,
                                                               float* __restrict__ out, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, k = i - n * D;
    float v = 0.f;
    const sgdfr_direction e = tab.d[k];
    if (which[n] == k && e.kind != SGDFR_DIR_ZERO) {
        const float start = place32(e, pick(src, e, n));
        const float lo = __fsub_rn(-shift_scale, start), hi = __fsub_rn(shift_scale, start);
        v = __fadd_rn(__fmul_rn(__fsub_rn(lo, hi), u[n]), hi);
    }
    out[i] = v;
}

static int check_table(const sgdfr_direction* table, int D, int pose_dim, int exp_dim, DirTable* out) {
    SGDFR_REQUIRE(table && D >= 1 && D <= SGDFR_MAX_DIRECTIONS, "make_shift: 1..%d directions, got %d", SGDFR_MAX_DIRECTIONS, D);
    for (int k = 0; k < D; ++k) {
        const sgdfr_direction& e = table[k];
        const int lim = e.kind == SGDFR_DIR_ANGLE ? 3 : e.kind == SGDFR_DIR_JAW ? pose_dim : e.kind == SGDFR_DIR_EXP ? exp_dim : 1;
        SGDFR_REQUIRE(e.kind >= SGDFR_DIR_ZERO && e.kind <= SGDFR_DIR_EXP, "make_shift: direction %d has unknown kind %d", k, e.kind);
        SGDFR_REQUIRE(e.col >= 0 && e.col < lim, "make_shift: direction %d reads column %d of %d", k, e.col, lim);
        SGDFR_REQUIRE(e.kind != SGDFR_DIR_ANGLE || e.b != 0.0, "make_shift: direction %d has a zero angle scale", k);
        out->d[k] = e;
    }
    return 0;
}

}  // namespace sgdfr

using namespace sgdfr;

extern "C" int sgdfr_make_shift_f32(const float* ang_s, int64_t ang_s_bs, const float* pose_s, int64_t pose_s_bs, const float* exp_s,
                                    int64_t exp_s_bs, const float* ang_t, const float* pose_t, const float* exp_t, int pose_dim,
                                    int exp_dim, const struct sgdfr_direction* table, int D, float* shift, int N, int arith,
                                    void* stream) {
    SGDFR_REQUIRE(N >= 0 && pose_dim >= 1 && exp_dim >= 1, "make_shift: bad sizes N=%d pose_dim=%d exp_dim=%d", N, pose_dim, exp_dim);
    SGDFR_REQUIRE(arith == 0 || arith == 1, "make_shift: arith must be 0 (float64 scalars) or 1 (float32 tensors), got %d", arith);
    DirTable tab;
    if (int rc = check_table(table, D, pose_dim, exp_dim, &tab)) return rc;
    if (N == 0) return 0;                  // empty batch: empty tensors carry null pointers
    SGDFR_REQUIRE(ang_s && pose_s && exp_s && ang_t && pose_t && exp_t && shift, "make_shift: null pointer");
    ShiftSrc src{{ang_s, pose_s, exp_s}, {ang_s_bs, pose_s_bs, exp_s_bs}};
    ShiftSrc tgt{{ang_t, pose_t, exp_t}, {3, pose_dim, exp_dim}};
    const int total = N * D;
    hipLaunchKernelGGL(make_shift_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), src, tgt, tab, D, shift, N, arith);
    return check_launch("make_shift");
}

extern "C" int sgdfr_make_shift_random_f32(const float* ang_s, const float* pose_s, const float* exp_s, int pose_dim, int exp_dim,
                                           const int* which, const float* u, float shift_scale,
                                           const struct sgdfr_direction* table, int D, float* shift, int N, void* stream) {
    SGDFR_REQUIRE(N >= 0 && pose_dim >= 1 && exp_dim >= 1, "make_shift_random: bad sizes N=%d pose_dim=%d exp_dim=%d", N, pose_dim,
                  exp_dim);
    DirTable tab;
    if (int rc = check_table(table, D, pose_dim, exp_dim, &tab)) return rc;
    if (N == 0) return 0;
    SGDFR_REQUIRE(ang_s && pose_s && exp_s && which && u && shift, "make_shift_random: null pointer");
    ShiftSrc src{{ang_s, pose_s, exp_s}, {3, pose_dim, exp_dim}};
    const int total = N * D;
    hipLaunchKernelGGL(make_shift_random_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), src, tab, D, which, u,
                       shift_scale, shift, N);
    return check_launch("make_shift_random");
}
