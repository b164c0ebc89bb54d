// Diffusion algebra, RNG and optimiser kernels — all fp32, HBM-bound streaming kernels.
// Reference: gms/diffusion/gaussian_diffusion.py (training_losses :81-172, _run_model :61-77, _cf_guidance :174-187,
// ddim_step :189-213, reverse_dpm_step :215-243, sample :292) and gms/diffusion/diffusion_utils.py
// (diffusion_forward :65-73, diffusion_reverse :34-62, predict_* :76-105, _logsnr_schedule_cosine :198-201);
// torch.optim.Adam as diffusion_model.py:56 constructs it.
#include <math.h>

#include "gmk_common.h"

namespace {

// cosine log-SNR schedule constants for logsnr in [-20, 20] (diffusion_utils.py:199-200), rounded to fp32 the way
// torch applies a Python/numpy double scalar to an fp32 tensor.
constexpr float kSchedB = 4.539992973129278e-05f;
constexpr float kSchedA = 1.5707055269354342f;

struct LogsnrCoef {
    float alpha, sigma;   // sqrt(sigmoid(l)), sqrt(sigmoid(-l))
    float c1, c2;         // sqrt(1 + e^l), rsqrt(1 + e^-l)   (predict_eps_from_x)
    float d1, d2;         // sqrt(1 + e^-l), rsqrt(1 + e^l)   (predict_x_from_eps)
};

__device__ __forceinline__ LogsnrCoef logsnr_coef(float l) {
    LogsnrCoef c;
    const float el = expf(l), eml = expf(-l);
    c.alpha = sqrtf(1.0f / (1.0f + eml));
    c.sigma = sqrtf(1.0f / (1.0f + el));
    c.c1 = sqrtf(1.0f + el);
    c.c2 = 1.0f / sqrtf(1.0f + eml);
    c.d1 = sqrtf(1.0f + eml);
    c.d2 = 1.0f / sqrtf(1.0f + el);
    return c;
}

__device__ __forceinline__ float clip1(float x) { return fminf(fmaxf(x, -1.0f), 1.0f); }

__device__ __forceinline__ float block_sum(float v, float* red) {   // 256 threads; every thread gets the total
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (ceil(n/1024), B): z = x*alpha + eps*sigma with logsnr = schedule(u[b])
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                      const float* __restrict__ u, float* __restrict__ logsnr,
                                                      float* __restrict__ z, int64_t n) {
    const int b = blockIdx.y;
    const float t = __fadd_rn(__fmul_rn(kSchedA, u[b]), kSchedB);
    const float l = -2.0f * logf(tanf(t));
    if (blockIdx.x == 0 && threadIdx.x == 0) logsnr[b] = l;
    const float alpha = sqrtf(1.0f / (1.0f + expf(-l)));
    const float sigma = sqrtf(1.0f / (1.0f + expf(l)));
    const int64_t base = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 3 < n) {
            float xv[4], ev[4], o[4];
            load4(x + base + i, xv);
            load4(eps + base + i, ev);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __fadd_rn(__fmul_rn(xv[k], alpha), __fmul_rn(sigma, ev[k]));
            store4(z + base + i, o);
        } else {
            for (int64_t j = i; j < n; ++j)
                z[base + j] = __fadd_rn(__fmul_rn(x[base + j], alpha), __fmul_rn(sigma, eps[base + j]));
        }
    }
}

// the network output parametrises x (gaussian_diffusion.py:58-73): mean_type 0 'v' (predict_x_from_v), 1 'eps'
// (predict_x_from_eps), 2 'x'; d x / d out is the second function
__device__ __forceinline__ float x_from_out(float out, float zz, const LogsnrCoef& c, int mt) {
    return mt == 0 ? c.alpha * zz - c.sigma * out : (mt == 1 ? c.d1 * (zz - out * c.d2) : out);
}
__device__ __forceinline__ float dx_dout(const LogsnrCoef& c, int mt) { return mt == 0 ? -c.sigma : (mt == 1 ? -c.d1 * c.d2 : 1.0f); }

// one block per sample: loss and (optionally) d loss / d (network output)
__global__ __launch_bounds__(256) void v_loss_kernel(const float* __restrict__ v, const float* __restrict__ z,
                                                    const float* __restrict__ x, const float* __restrict__ eps,
                                                    const float* __restrict__ logsnr, float* __restrict__ loss_b,
                                                    float* __restrict__ x_mse_o, float* __restrict__ eps_mse_o,
                                                    float* __restrict__ dv, float grad_scale, int64_t n, int loss_type, int mt) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const LogsnrCoef c = logsnr_coef(logsnr[b]);
    const int64_t base = (int64_t)b * n;
    float sx = 0.f, se = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float zz = z[base + i];
        const float xh = clip1(x_from_out(v[base + i], zz, c, mt));        // :58-63, :73
        const float eh = c.c1 * (zz - xh * c.c2);                          // :76
        const float dx = xh - x[base + i], de = eh - eps[base + i];
        sx = fmaf(dx, dx, sx);
        se = fmaf(de, de, se);
    }
    sx = block_sum(sx, red);
    se = block_sum(se, red);
    const float xm = sx / (float)n, em = se / (float)n;
    if (threadIdx.x == 0) {
        loss_b[b] = loss_type == 1 ? em : fmaxf(xm, em);                   // :171 'snr' (distillation step1) / :169 'snr_trunc'
        if (x_mse_o) x_mse_o[b] = xm;
        if (eps_mse_o) eps_mse_o[b] = em;
    }
    if (!dv) return;
    // torch.maximum routes the gradient to the larger branch, ties split evenly; torch.clip passes it inside [-1, 1]
    const float gx = loss_type == 1 ? 0.f : (xm > em ? 1.f : (xm == em ? 0.5f : 0.f));
    const float ge = 1.f - gx;
    const float kx = grad_scale * gx * 2.f / (float)n;
    const float ke = grad_scale * ge * 2.f / (float)n * (-c.c1 * c.c2);
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const float zz = z[base + i];
        const float raw = x_from_out(v[base + i], zz, c, mt);
        const float xh = clip1(raw);
        const float eh = c.c1 * (zz - xh * c.c2);
        const float dxh = kx * (xh - x[base + i]) + ke * (eh - eps[base + i]);
        dv[base + i] = (raw >= -1.0f && raw <= 1.0f) ? dx_dout(c, mt) * dxh : 0.f;
    }
}

// grid (ceil(n/256), B)
__global__ __launch_bounds__(256) void sampler_step_kernel(const float* __restrict__ v, const float* __restrict__ vu,
                                                          const float* __restrict__ cond_w, const float* __restrict__ z,
                                                          const float* __restrict__ noise, float lt, float ls, int is_last,
                                                          float* __restrict__ z_next, float* __restrict__ x_pred,
                                                          float* __restrict__ eps_pred, int64_t n,
                                                          const float* __restrict__ lt_vec, const float* __restrict__ ls_vec, int mt,
                                                          float* __restrict__ z_dup = nullptr, float* __restrict__ logsnr_next = nullptr) {
    const int b = blockIdx.y;
    if (lt_vec) { lt = lt_vec[b]; ls = ls_vec[b]; }      // per-sample times (teacher steps of the distillation loss)
    // the next iteration's network time: u_t(i - 1) = u_s(i) (gaussian_diffusion.py:288-290), so logsnr_s IS the next logsnr_t
    if (logsnr_next && blockIdx.x == 0 && threadIdx.x == 0) {
        logsnr_next[b] = ls;
        if (z_dup) logsnr_next[gridDim.y + b] = ls;
    }
    const LogsnrCoef c = logsnr_coef(lt);
    const float alpha_s = sqrtf(1.0f / (1.0f + expf(-ls)));
    const float sigma_s = sqrtf(1.0f / (1.0f + expf(ls)));
    // ancestral posterior q(z_s | z_t, x) with x_logvar = 'large' (diffusion_utils.py:36-50)
    const float alpha_st = sqrtf((1.0f + expf(-lt)) / (1.0f + expf(-ls)));
    const float r = expf(lt - ls);
    const float omr = -expm1f(lt - ls);
    const float stdv = sqrtf(omr * (1.0f / (1.0f + expf(lt))));
    const float w = cond_w ? cond_w[b] : 0.f;
    const int64_t base = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float zz = z[base + i];
        float xh = clip1(x_from_out(v[base + i], zz, c, mt));
        float eh = c.c1 * (zz - xh * c.c2);
        if (vu) {   // classifier-free guidance in eps space (:176-186)
            const float xu = clip1(x_from_out(vu[base + i], zz, c, mt));
            const float eu = c.c1 * (zz - xu * c.c2);
            const float e = (1.0f + w) * eh + (-w) * eu;
            xh = clip1(c.d1 * (zz - e * c.d2));
            eh = c.c1 * (zz - xh * c.c2);
        }
        float zs;
        if (noise) zs = (r * alpha_st * zz + omr * alpha_s * xh) + stdv * noise[base + i];   // :242
        else zs = alpha_s * xh + sigma_s * eh;                                                 // :212
        z_next[base + i] = is_last ? xh : zs;                                                  // :292
        if (z_dup) z_dup[base + i] = is_last ? xh : zs;      // second half of the guided sampler's 2B-image batch (cond + uncond share z)
        if (x_pred) x_pred[base + i] = xh;
        if (eps_pred) eps_pred[base + i] = eh;
    }
}

template <bool NORMAL>
__global__ __launch_bounds__(256) void rng_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
    const int64_t nq = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t rnd[4];
        philox4x32(offset + (uint64_t)q, seed, rnd);
        float o[4];
        if (NORMAL) {   // Box-Muller on two pairs
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float u1 = 1.0f - u01(rnd[2 * k]);          // (0, 1]
                const float u2 = u01(rnd[2 * k + 1]);
                const float rad = sqrtf(-2.0f * logf(u1));
                float s, cs;
                sincosf(6.283185307179586f * u2, &s, &cs);
                o[2 * k] = rad * cs; o[2 * k + 1] = rad * s;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = u01(rnd[k]);
        }
        const int64_t i = q * 4;
        if (i + 3 < n) store4(out + i, o);
        else for (int k = 0; i + k < n; ++k) out[i + k] = o[k];
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, float step_size, float beta1,
                                                  float beta2, float eps, float inv_bc2_sqrt, float grad_scale) {
    const int64_t nq = n / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        float pv[4], gv[4], mv[4], vv[4];
        load4(p + q * 4, pv); load4(g + q * 4, gv); load4(m + q * 4, mv); load4(v + q * 4, vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gg = gv[k] * grad_scale;
            mv[k] = mv[k] + (gg - mv[k]) * (1.0f - beta1);             // exp_avg.lerp_(grad, 1 - beta1)
            vv[k] = vv[k] * beta2 + (1.0f - beta2) * gg * gg;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
            const float denom = sqrtf(vv[k]) * inv_bc2_sqrt + eps;
            pv[k] = pv[k] - step_size * (mv[k] / denom);
        }
        store4(p + q * 4, pv); store4(m + q * 4, mv); store4(v + q * 4, vv);
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = nq * 4 + threadIdx.x;
        const float gg = g[i] * grad_scale;
        const float mm = m[i] + (gg - m[i]) * (1.0f - beta1);
        const float vv = v[i] * beta2 + (1.0f - beta2) * gg * gg;
        m[i] = mm; v[i] = vv;
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
    }
}

int stream_grid(int64_t work_items) {
    int64_t g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int gmk_rng_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    GMK_REQUIRE(out && n > 0, "gmk_rng_normal: bad arguments");
    rng_kernel<true><<<stream_grid((n + 3) / 4), 256, 0, gmk_stream(stream)>>>(out, n, seed, offset);
    return gmk_check_launch("gmk_rng_normal");
}

extern "C" int gmk_rng_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    GMK_REQUIRE(out && n > 0, "gmk_rng_uniform: bad arguments");
    rng_kernel<false><<<stream_grid((n + 3) / 4), 256, 0, gmk_stream(stream)>>>(out, n, seed, offset);
    return gmk_check_launch("gmk_rng_uniform");
}

extern "C" int gmk_q_sample(const float* x, const float* eps, const float* u, float* logsnr, float* z, int B, int64_t n,
                            void* stream) {
    GMK_REQUIRE(x && eps && u && logsnr && z, "gmk_q_sample: null pointer");
    GMK_REQUIRE(B > 0 && B < 65536 && n > 0 && n % 4 == 0, "gmk_q_sample: bad shape B=%d n=%lld (n %% 4 == 0 required)", B,
                (long long)n);
    int gx = (int)((n + 1023) / 1024);
    if (gx > 64) gx = 64;
    q_sample_kernel<<<dim3(gx, B), 256, 0, gmk_stream(stream)>>>(x, eps, u, logsnr, z, n);
    return gmk_check_launch("gmk_q_sample");
}

extern "C" int gmk_v_loss(const float* v, const float* z, const float* x, const float* eps, const float* logsnr,
                          float* loss_b, float* x_mse, float* eps_mse, float* dv, float grad_scale, int loss_type,
                          int mean_type, int B, int64_t n, void* stream) {
    GMK_REQUIRE(v && z && x && eps && logsnr && loss_b, "gmk_v_loss: null pointer");
    GMK_REQUIRE(B > 0 && n > 0, "gmk_v_loss: bad shape");
    GMK_REQUIRE(loss_type == 0 || loss_type == 1, "gmk_v_loss: loss_type must be 0 (snr_trunc) or 1 (snr)");
    GMK_REQUIRE(mean_type >= 0 && mean_type <= 2, "gmk_v_loss: mean_type must be 0 (v), 1 (eps) or 2 (x)");
    v_loss_kernel<<<B, 256, 0, gmk_stream(stream)>>>(v, z, x, eps, logsnr, loss_b, x_mse, eps_mse, dv, grad_scale, n, loss_type, mean_type);
    return gmk_check_launch("gmk_v_loss");
}

extern "C" int gmk_sampler_step(const float* v, const float* v_uncond, const float* cond_w, const float* z,
                                const float* noise, float logsnr_t, float logsnr_s, int is_last, float* z_next,
                                float* x_pred, float* eps_pred, float* z_dup, float* logsnr_next, int mean_type, int B, int64_t n,
                                void* stream) {
    GMK_REQUIRE(v && z && z_next, "gmk_sampler_step: null pointer");
    GMK_REQUIRE(mean_type >= 0 && mean_type <= 2, "gmk_sampler_step: mean_type must be 0 (v), 1 (eps) or 2 (x)");
    GMK_REQUIRE((v_uncond == nullptr) == (cond_w == nullptr), "gmk_sampler_step: v_uncond and cond_w go together");
    GMK_REQUIRE(B > 0 && B < 65536 && n > 0, "gmk_sampler_step: bad shape");
    int gx = (int)((n + 255) / 256);
    if (gx > 64) gx = 64;
    sampler_step_kernel<<<dim3(gx, B), 256, 0, gmk_stream(stream)>>>(v, v_uncond, cond_w, z, noise, logsnr_t, logsnr_s,
                                                                     is_last, z_next, x_pred, eps_pred, n, nullptr, nullptr, mean_type,
                                                                     z_dup, logsnr_next);
    return gmk_check_launch("gmk_sampler_step");
}

extern "C" int gmk_ddim_step_vec(const float* v, const float* v_uncond, const float* cond_w, const float* z,
                                 const float* logsnr_t, const float* logsnr_s, float* z_next, float* x_pred, float* eps_pred,
                                 int mean_type, int B, int64_t n, void* stream) {
    GMK_REQUIRE(v && z && z_next && logsnr_t && logsnr_s, "gmk_ddim_step_vec: null pointer");
    GMK_REQUIRE(mean_type >= 0 && mean_type <= 2, "gmk_ddim_step_vec: mean_type must be 0 (v), 1 (eps) or 2 (x)");
    GMK_REQUIRE((v_uncond == nullptr) == (cond_w == nullptr), "gmk_ddim_step_vec: v_uncond and cond_w go together");
    GMK_REQUIRE(B > 0 && B < 65536 && n > 0, "gmk_ddim_step_vec: bad shape");
    int gx = (int)((n + 255) / 256);
    if (gx > 64) gx = 64;
    sampler_step_kernel<<<dim3(gx, B), 256, 0, gmk_stream(stream)>>>(v, v_uncond, cond_w, z, nullptr, 0.f, 0.f, 0, z_next, x_pred,
                                                                     eps_pred, n, logsnr_t, logsnr_s, mean_type);
    return gmk_check_launch("gmk_ddim_step_vec");
}

namespace {
// u -> logsnr (diffusion_utils.py:198-201); optionally u = (i + 1) / T - shift from integer times (gaussian_diffusion.py:90-91)
__global__ void schedule_kernel(const float* __restrict__ u, const int64_t* __restrict__ ti, float inv_T_num, float T,
                                float shift, float* __restrict__ u_out, float* __restrict__ logsnr, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float uu = ti ? __fdiv_rn((float)(ti[b] + 1), T) : u[b];
    (void)inv_T_num;
    uu = __fsub_rn(uu, shift);
    if (u_out) u_out[b] = uu;
    const float t = __fadd_rn(__fmul_rn(kSchedA, uu), kSchedB);
    logsnr[b] = -2.0f * logf(tanf(t));
}

// gaussian_diffusion.py:147-154: x-target implied by two teacher DDIM steps, its i == 0 select, and the eps-target
__global__ __launch_bounds__(256) void distill_target_kernel(const float* __restrict__ z_teacher, const float* __restrict__ z_t,
                                                            const float* __restrict__ x_pred_teacher,
                                                            const float* __restrict__ logsnr, const float* __restrict__ logsnr_s,
                                                            const int64_t* __restrict__ ti, float* __restrict__ x_target,
                                                            float* __restrict__ eps_target, int64_t n) {
    const int b = blockIdx.y;
    const float l = logsnr[b], ls = logsnr_s[b];
    const float alpha_s = sqrtf(1.0f / (1.0f + expf(-ls)));
    const float alpha_t = sqrtf(1.0f / (1.0f + expf(-l)));
    // F.softplus(x) = log1p(exp(x)) (x <= 20), x beyond the threshold
    const float sp_t = l > 20.f ? l : log1pf(expf(l));
    const float sp_s = ls > 20.f ? ls : log1pf(expf(ls));
    const float frac = expf(0.5f * (sp_t - sp_s));
    const float denom = alpha_s - frac * alpha_t;
    const float c1 = sqrtf(1.0f + expf(l)), c2 = 1.0f / sqrtf(1.0f + expf(-l));
    const bool first = ti[b] == 0;
    const int64_t base = (int64_t)b * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float zt = z_t[base + i];
        const float xt = first ? x_pred_teacher[base + i] : (z_teacher[base + i] - frac * zt) / denom;
        x_target[base + i] = xt;
        eps_target[base + i] = c1 * (zt - xt * c2);
    }
}
}  // namespace

extern "C" int gmk_logsnr_schedule(const float* u, const int64_t* i_times, int num_steps, float shift, float* u_out,
                                   float* logsnr, int B, void* stream) {
    GMK_REQUIRE((u != nullptr) != (i_times != nullptr) && logsnr && B > 0, "gmk_logsnr_schedule: give exactly one of u / i_times");
    GMK_REQUIRE(!i_times || num_steps >= 1, "gmk_logsnr_schedule: num_steps");
    schedule_kernel<<<(B + 255) / 256, 256, 0, gmk_stream(stream)>>>(u, i_times, 0.f, (float)num_steps, shift, u_out, logsnr, B);
    return gmk_check_launch("gmk_logsnr_schedule");
}

extern "C" int gmk_distill_target(const float* z_teacher, const float* z_t, const float* x_pred_teacher, const float* logsnr,
                                  const float* logsnr_s, const int64_t* i_times, float* x_target, float* eps_target, int B,
                                  int64_t n, void* stream) {
    GMK_REQUIRE(z_teacher && z_t && x_pred_teacher && logsnr && logsnr_s && i_times && x_target && eps_target,
                "gmk_distill_target: null pointer");
    GMK_REQUIRE(B > 0 && B < 65536 && n > 0, "gmk_distill_target: bad shape");
    int gx = (int)((n + 255) / 256);
    if (gx > 64) gx = 64;
    distill_target_kernel<<<dim3(gx, B), 256, 0, gmk_stream(stream)>>>(z_teacher, z_t, x_pred_teacher, logsnr, logsnr_s, i_times,
                                                                       x_target, eps_target, n);
    return gmk_check_launch("gmk_distill_target");
}

extern "C" int gmk_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, void* stream) {
    GMK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "gmk_adam_step: bad arguments");
    // scalar prologue in double, as torch.optim.adam._single_tensor_adam does on the host
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    adam_kernel<<<stream_grid(n / 4 + 1), 256, 0, gmk_stream(stream)>>>(p, g, m, v, n, step_size, beta1, beta2, eps,
                                                                        inv_bc2_sqrt, grad_scale);
    return gmk_check_launch("gmk_adam_step");
}
