// Error plumbing + the HBM-bound pointwise pieces of the generator path:
//   fused bias + leaky-ReLU (+ its first-order grad form), PixelNorm, latent preparation.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace sgdfr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

// ---------------------------------------------------------------- fused_bias_act
// 4 elements per thread when the bias period allows a float4 (step_b % 4 == 0 or no bias):
// 16 B/lane coalesced, grid capped and grid-strided.
template <bool VEC>
__global__ __launch_bounds__(256) void fused_bias_act_kernel(const float* __restrict__ x, const float* __restrict__ bias,
                                                            const float* __restrict__ ref, float* __restrict__ y,
                                                            int64_t n, int step_b, int size_b, int grad, float alpha,
                                                            float scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            float b = bias ? bias[((i << 2) / step_b) % size_b] : 0.f;
            float4 r = ref ? reinterpret_cast<const float4*>(ref)[i] : v;
            float4 o;
            if (grad == 0) {
                v.x += b; v.y += b; v.z += b; v.w += b;
                o.x = (v.x > 0.f ? v.x : v.x * alpha) * scale;
                o.y = (v.y > 0.f ? v.y : v.y * alpha) * scale;
                o.z = (v.z > 0.f ? v.z : v.z * alpha) * scale;
                o.w = (v.w > 0.f ? v.w : v.w * alpha) * scale;
            } else if (grad == 1) {
                o.x = (r.x > 0.f ? v.x : v.x * alpha) * scale;
                o.y = (r.y > 0.f ? v.y : v.y * alpha) * scale;
                o.z = (r.z > 0.f ? v.z : v.z * alpha) * scale;
                o.w = (r.w > 0.f ? v.w : v.w * alpha) * scale;
            } else {
                o = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            reinterpret_cast<float4*>(y)[i] = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            float v = x[i];
            float b = bias ? bias[(i / step_b) % size_b] : 0.f;
            float r = ref ? ref[i] : 0.f;
            float o;
            if (grad == 0) {
                v += b;
                o = (v > 0.f ? v : v * alpha) * scale;
            } else if (grad == 1) {
                o = (r > 0.f ? v : v * alpha) * scale;
            } else {
                o = 0.f;
            }
            y[i] = o;
        }
    }
}

// The other two dtypes the reference's native dispatches (AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79): one
// element per thread, arithmetic in float (half) / double (double), result rounded once.
template <typename T, typename A>
__global__ __launch_bounds__(256) void fused_bias_act_typed_kernel(const T* __restrict__ x, const T* __restrict__ bias,
                                                                  const T* __restrict__ ref, T* __restrict__ y, int64_t n,
                                                                  int step_b, int size_b, int grad, A alpha, A scale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        A v = (A)x[i];
        const A b = bias ? (A)bias[(i / step_b) % size_b] : (A)0;
        const A r = ref ? (A)ref[i] : (A)0;
        A o;
        if (grad == 0) {
            v += b;
            o = (v > (A)0 ? v : v * alpha) * scale;
        } else if (grad == 1) {
            o = (r > (A)0 ? v : v * alpha) * scale;
        } else {
            o = (A)0;
        }
        y[i] = (T)o;
    }
}

// ---------------------------------------------------------------- pixelnorm
// one wave per row: wave-shuffle reduction over D
__global__ __launch_bounds__(256) void pixelnorm_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D,
                                                       float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (int64_t)row * D;
    float acc = 0.f;
    for (int j = lane; j < D; j += 64) acc += xr[j] * xr[j];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const float r = rsqrtf(acc / (float)D + eps);
    for (int j = lane; j < D; j += 64) y[(int64_t)row * D + j] = xr[j] * r;
}

// adjoint of pixelnorm: y = x*r, r = rsqrt(mean x^2 + eps)  =>  dx = r*g - x * r^3 * mean(g*x); one wave per row
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           float* __restrict__ dx, int B, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float* xr = x + (int64_t)row * D;
    const float* gr = g + (int64_t)row * D;
    float sq = 0.f, gx = 0.f;
    for (int j = lane; j < D; j += 64) {
        sq = fmaf(xr[j], xr[j], sq);
        gx = fmaf(gr[j], xr[j], gx);
    }
    for (int o = 32; o > 0; o >>= 1) {
        sq += __shfl_xor(sq, o, 64);
        gx += __shfl_xor(gx, o, 64);
    }
    const float r = rsqrtf(sq / (float)D + eps);
    const float c = r * r * r * (gx / (float)D);
    for (int j = lane; j < D; j += 64) dx[(int64_t)row * D + j] = r * gr[j] - xr[j] * c;
}

// ---------------------------------------------------------------- latent prepare
__global__ __launch_bounds__(256) void latent_prepare_kernel(const float* __restrict__ w, int w_is_plus,
                                                            const float* __restrict__ shift, int shift_is_plus,
                                                            int shift_layers, const float* __restrict__ trunc, float psi,
                                                            float* __restrict__ out, int B, int L, int D) {
    const int64_t n = (int64_t)B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % D);
        const int l = (int)((i / D) % L);
        const int b = (int)(i / ((int64_t)D * L));
        float v = w_is_plus ? w[i] : w[(int64_t)b * D + j];
        if (shift && l < shift_layers)
            v += shift_is_plus ? shift[((int64_t)b * shift_layers + l) * D + j] : shift[(int64_t)b * D + j];
        if (trunc) {
            const float t = trunc[j];
            v = t + psi * (v - t);
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------- output conversion
// [B,3,H,W] fp32 in [-1,1] -> [B,H,W,3] uint8 with the reference's scaling
// (libs/utilities/image_utils.py:87-110: clamp to [-1,1], (v+1)/(2+1e-5)*255, then the uint8 truncation of the writers)
__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ x, unsigned char* __restrict__ y, int B,
                                                         int HW) {
    const int64_t n = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW;
        const int p = (int)(i - b * HW);
        const float* xp = x + b * 3 * HW + p;
        unsigned char* yp = y + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = fminf(fmaxf(xp[(int64_t)c * HW], -1.f), 1.f);
            v = (v + 1.f) / (2.f + 1e-5f) * 255.f;
            yp[c] = (unsigned char)v;
        }
    }
}

// K panels [B,3,H,W] (batch stride 0 = one image shown in every row) side by side -> video frames [B,H,K*W,3] uint8:
// generate_grid_image + tensor_to_image + the np.uint8 of generate_video (libs/utilities/utils_inference.py:11-33,
// run_inference.py:188-194) for a whole batch of frames.  swap_rb: the cvtColor(.., COLOR_BGR2RGB) before VideoWriter.
struct GridPanels {
    const float* x[SGDFR_MAX_GRID_PANELS];
    int64_t bstride[SGDFR_MAX_GRID_PANELS];
};

__global__ __launch_bounds__(256) void grid_to_u8_kernel(GridPanels g, unsigned char* __restrict__ y, int B, int H, int W,
                                                        int K, int swap_rb) {
    const int64_t n = (int64_t)B * H * K * W;
    const int HW = H * W, KW = K * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % KW);
        const int64_t br = i / KW;
        const int row = (int)(br % H);
        const int64_t b = br / H;
        const int k = col / W, c0 = col - k * W;
        if (!g.x[k]) continue;                      // skipped panel: filled by the fused ToRGB finish
        const float* xp = g.x[k] + b * g.bstride[k] + row * W + c0;
        unsigned char* yp = y + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = fminf(fmaxf(xp[(int64_t)c * HW], -1.f), 1.f);
            v = (v + 1.f) / (2.f + 1e-5f) * 255.f;
            yp[swap_rb ? 2 - c : c] = (unsigned char)v;
        }
    }
}

}  // namespace sgdfr

using namespace sgdfr;

extern "C" int sgdfr_abi_version(void) { return SGDFR_ABI_VERSION; }
extern "C" const char* sgdfr_last_error(void) { return g_err; }

static int grid_for(int64_t work_items, int cap = 256 * 8) {
    int64_t g = (work_items + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

extern "C" int sgdfr_fused_bias_act_f32(const float* x, const float* bias, const float* ref, float* y, int64_t n,
                                        int step_b, int size_b, int act, int grad, float alpha, float scale,
                                        void* stream) {
    SGDFR_REQUIRE(act == 3, "fused_bias_act: only act=3 (leaky relu) is implemented, got %d", act);
    SGDFR_REQUIRE(grad >= 0 && grad <= 2, "fused_bias_act: grad must be 0,1,2, got %d", grad);
    SGDFR_REQUIRE(n >= 0, "fused_bias_act: negative n");
    if (n == 0) return 0;
    SGDFR_REQUIRE(x && y, "fused_bias_act: null x/y");
    SGDFR_REQUIRE(!bias || (step_b > 0 && size_b > 0), "fused_bias_act: bad bias geometry %d %d", step_b, size_b);
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                           (ref ? reinterpret_cast<uintptr_t>(ref) : 0)) & 15) == 0;
    const bool vec = aligned && (n % 4 == 0) && (!bias || step_b % 4 == 0);
    if (vec)
        hipLaunchKernelGGL(fused_bias_act_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, as_stream(stream), x, bias,
                           ref, y, n, step_b, size_b, grad, alpha, scale);
    else
        hipLaunchKernelGGL(fused_bias_act_kernel<false>, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, bias,
                           ref, y, n, step_b, size_b, grad, alpha, scale);
    return check_launch("fused_bias_act");
}

extern "C" int sgdfr_fused_bias_act(const void* x, const void* bias, const void* ref, void* y, int64_t n, int step_b, int size_b,
                                    int act, int grad, float alpha, float scale, int dtype, void* stream) {
    if (dtype == SGDFR_DTYPE_F32)
        return sgdfr_fused_bias_act_f32(static_cast<const float*>(x), static_cast<const float*>(bias), static_cast<const float*>(ref),
                                        static_cast<float*>(y), n, step_b, size_b, act, grad, alpha, scale, stream);
    SGDFR_REQUIRE(dtype == SGDFR_DTYPE_F16 || dtype == SGDFR_DTYPE_F64, "fused_bias_act: dtype must be SGDFR_DTYPE_F32/F16/F64, got %d", dtype);
    SGDFR_REQUIRE(act == 3, "fused_bias_act: only act=3 (leaky relu) is implemented, got %d", act);
    SGDFR_REQUIRE(grad >= 0 && grad <= 2, "fused_bias_act: grad must be 0,1,2, got %d", grad);
    SGDFR_REQUIRE(n >= 0, "fused_bias_act: negative n");
    if (n == 0) return 0;
    SGDFR_REQUIRE(x && y, "fused_bias_act: null x/y");
    SGDFR_REQUIRE(!bias || (step_b > 0 && size_b > 0), "fused_bias_act: bad bias geometry %d %d", step_b, size_b);
    if (dtype == SGDFR_DTYPE_F16)
        hipLaunchKernelGGL((fused_bias_act_typed_kernel<_Float16, float>), dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                           static_cast<const _Float16*>(x), static_cast<const _Float16*>(bias), static_cast<const _Float16*>(ref),
                           static_cast<_Float16*>(y), n, step_b, size_b, grad, alpha, scale);
    else
        hipLaunchKernelGGL((fused_bias_act_typed_kernel<double, double>), dim3(grid_for(n)), dim3(256), 0, as_stream(stream),
                           static_cast<const double*>(x), static_cast<const double*>(bias), static_cast<const double*>(ref),
                           static_cast<double*>(y), n, step_b, size_b, grad, (double)alpha, (double)scale);
    return check_launch("fused_bias_act");
}

extern "C" int sgdfr_pixelnorm_f32(const float* x, float* y, int B, int D, float eps, void* stream) {
    SGDFR_REQUIRE(B >= 0 && D > 0, "pixelnorm: bad shape %d %d", B, D);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && y, "pixelnorm: null pointer");
    hipLaunchKernelGGL(pixelnorm_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), x, y, B, D, eps);
    return check_launch("pixelnorm");
}

extern "C" int sgdfr_pixelnorm_bwd_f32(const float* x, const float* g, float* dx, int B, int D, float eps, void* stream) {
    SGDFR_REQUIRE(B >= 0 && D > 0, "pixelnorm_bwd: bad shape B=%d D=%d", B, D);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && g && dx, "pixelnorm_bwd: null pointer");
    hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, as_stream(stream), x, g, dx, B, D, eps);
    return check_launch("pixelnorm_bwd");
}

extern "C" int sgdfr_latent_prepare_f32(const float* w, int w_is_plus, const float* shift, int shift_is_plus,
                                        int shift_layers, const float* trunc, float psi, float* out, int B, int L,
                                        int D, void* stream) {
    SGDFR_REQUIRE(B >= 0 && L > 0 && D > 0, "latent_prepare: bad shape %d %d %d", B, L, D);
    if (B == 0) return 0;
    SGDFR_REQUIRE(w && out, "latent_prepare: null pointer");
    SGDFR_REQUIRE(!shift || (shift_layers >= 0 && shift_layers <= L), "latent_prepare: shift_layers %d > L %d",
                  shift_layers, L);
    hipLaunchKernelGGL(latent_prepare_kernel, dim3(grid_for((int64_t)B * L * D)), dim3(256), 0, as_stream(stream), w,
                       w_is_plus, shift, shift_is_plus, shift_layers, trunc, psi, out, B, L, D);
    return check_launch("latent_prepare");
}

// ---------------------------------------------------------------- Adam over a list of parameter tensors, one launch
// The PTI step (libs/optimization.py:41,66-68: torch.optim.Adam, default betas / eps, no weight decay) updates 24 parameter tensors;
// torch's capturable multi-tensor Adam takes ~100 launches for them under a hipGraph (its per-parameter step-size tensors).  Here
// blockIdx -> (tensor, 1024-element chunk); the step count lives on the device (the captured step increments it before this launch).
//   m = m + (g - m) (1 - b1) ; v = v b2 + (1 - b2) g^2 ; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// (torch/optim/adam.py _single_tensor_adam, the non-capturable order of operations)
struct AdamBatch {
    sgdfr_adam_tensor t[SGDFR_MAX_ADAM_TENSORS];
    int chunk_start[SGDFR_MAX_ADAM_TENSORS + 1];
    int n;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamBatch ab, const float* __restrict__ step, float lr, float b1, float b2, float eps) {
    int ti = 0;
    while (ti + 1 < ab.n && (int)blockIdx.x >= ab.chunk_start[ti + 1]) ++ti;
    const sgdfr_adam_tensor& e = ab.t[ti];
    const float t = step[0];
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    const int64_t i0 = (int64_t)(blockIdx.x - ab.chunk_start[ti]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * 256 + threadIdx.x;
        if (i >= e.n) break;
        const float g = e.g[i];
        float m = e.m[i], v = e.v[i];
        m = m + (g - m) * (1.f - b1);
        v = fmaf(g * g, 1.f - b2, v * b2);
        e.m[i] = m;
        e.v[i] = v;
        e.p[i] = e.p[i] - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}

extern "C" int sgdfr_adam_f32(const sgdfr_adam_tensor* tensors, int n, const float* step, float lr, float beta1, float beta2, float eps,
                              void* stream) {
    SGDFR_REQUIRE(n > 0 && n <= SGDFR_MAX_ADAM_TENSORS && tensors && step, "adam: 1..%d tensors and a device step count", SGDFR_MAX_ADAM_TENSORS);
    AdamBatch ab;
    ab.n = n;
    int chunks = 0;
    for (int i = 0; i < n; ++i) {
        SGDFR_REQUIRE(tensors[i].p && tensors[i].g && tensors[i].m && tensors[i].v && tensors[i].n > 0, "adam: tensor %d: null pointer / empty", i);
        ab.t[i] = tensors[i];
        ab.chunk_start[i] = chunks;
        chunks += (int)((tensors[i].n + 1023) / 1024);
    }
    ab.chunk_start[n] = chunks;
    hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(256), 0, as_stream(stream), ab, step, lr, beta1, beta2, eps);
    return check_launch("adam");
}

extern "C" int sgdfr_image_to_u8_f32(const float* x, unsigned char* y, int B, int H, int W, void* stream) {
    SGDFR_REQUIRE(B >= 0 && H > 0 && W > 0, "image_to_u8: bad shape %d %d %d", B, H, W);
    if (B == 0) return 0;
    SGDFR_REQUIRE(x && y, "image_to_u8: null pointer");
    hipLaunchKernelGGL(image_to_u8_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, as_stream(stream), x, y, B,
                       H * W);
    return check_launch("image_to_u8");
}

extern "C" int sgdfr_grid_to_u8_f32(const float* const* panels, const int64_t* bstrides, int K, unsigned char* y, int B,
                                    int H, int W, int swap_rb, void* stream) {
    SGDFR_REQUIRE(B >= 0 && H > 0 && W > 0, "grid_to_u8: bad shape %d %d %d", B, H, W);
    SGDFR_REQUIRE(K >= 1 && K <= SGDFR_MAX_GRID_PANELS, "grid_to_u8: %d panels (1..%d)", K, SGDFR_MAX_GRID_PANELS);
    if (B == 0) return 0;
    SGDFR_REQUIRE(panels && bstrides && y, "grid_to_u8: null pointer");
    GridPanels g;
    for (int k = 0; k < SGDFR_MAX_GRID_PANELS; ++k) {
        g.x[k] = k < K ? panels[k] : nullptr;
        g.bstride[k] = k < K ? bstrides[k] : 0;
        // a null panel is SKIPPED: its columns of y keep what sgdfr_torgb_finish_u8_f32 already wrote there
    }
    hipLaunchKernelGGL(grid_to_u8_kernel, dim3(grid_for((int64_t)B * H * K * W)), dim3(256), 0, as_stream(stream), g, y,
                       B, H, W, K, swap_rb);
    return check_launch("grid_to_u8");
}
