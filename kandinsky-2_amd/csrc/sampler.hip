// k22 — fused improved-DDPM ancestral sampler step ("p_sampler"), device side, no host sync.
//
// Restates, for one step (file:line relative to /root/reference):
//   classifier-free guidance        kandinsky2/kandinsky2_1_model.py:222-233  (model_fn)
//   learned-range variance          kandinsky2/model/gaussian_diffusion.py:253-267
//   eps -> x0                       gaussian_diffusion.py:324-329
//   denoised_fn clamp / inpaint mix kandinsky2_1_model.py:237-243
//   dynamic threshold               gaussian_diffusion.py:284-294  (np.percentile(|x0|, 99.5) of batch
//                                   element 0 ONLY, linear interpolation, applied to the whole batch)
//   posterior mean, sample          gaussian_diffusion.py:189-207, 377-381
// The reference copies x0 to the host and sorts it there every step (gaussian_diffusion.py:288-290);
// here the two order statistics come from an exact 4x8-bit radix select on the fp32 bit patterns.
//
// table[step][8] (fp32, from the fp64 host tables exactly like _extract_into_tensor(...).float()):
//   0 sqrt_recip_alphas_cumprod   1 sqrt_recipm1_alphas_cumprod   2 posterior_mean_coef1
//   3 posterior_mean_coef2        4 posterior_log_variance_clipped 5 log(beta)
//   6 (t != 0)                    7 model timestep (after respacing map and 1000/T rescale)
#include "kernels.h"
#include "elementwise.h"

__device__ __forceinline__ const float* step_row(const SamplerParams& p) {
  const int st = p.step ? *p.step : p.step_host;
  return p.table + (int64_t)st * 8;
}

__device__ __forceinline__ float guided_eps(const SamplerParams& p, int n, int c, int pix) {
  const int64_t hw = p.HW;
  if (!p.use_cfg) return p.model_out[((int64_t)n * 8 + c) * hw + pix];
  const int bs = p.N / 2, i = n % bs;
  const float ce = p.model_out[((int64_t)i * 8 + c) * hw + pix];
  const float ue = p.model_out[((int64_t)(i + bs) * 8 + c) * hw + pix];
  return ue + p.guidance * (ce - ue);
}

// pass 1: x0 = denoised_fn(sqrt_recip * x - sqrt_recipm1 * eps)
__global__ __launch_bounds__(256) void sampler_x0_kernel(SamplerParams p) {
  const float* tr = step_row(p);
  const float sr = tr[0], srm1 = tr[1];
  const int64_t total = (int64_t)p.N * 4 * p.HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % p.HW);
    const int c = (int)((i / p.HW) % 4), n = (int)(i / (4 * (int64_t)p.HW));
    const float eps = guided_eps(p, n, c, pix);
    float x0 = __fsub_rn(__fmul_rn(sr, p.x[i]), __fmul_rn(srm1, eps));
    x0 = fminf(fmaxf(x0, p.clamp_lo), p.clamp_hi);
    if (p.mask != nullptr) {
      const float mk = p.mask[(int64_t)n * p.HW + pix];
      x0 = __fadd_rn(__fmul_rn(x0, 1.f - mk), __fmul_rn(p.init_img[i], mk));
    }
    p.x0_buf[i] = x0;
  }
}

// pass 2: s = max(percentile_99.5(|x0[0]|), 1) — exact order statistics by radix select.
// One 1024-thread workgroup.  |x0| is clamped, so the high key bytes collapse into two or three histogram
// bins: lanes of a wave that hit the same bin are combined with ballots and ONE LDS atomic per distinct bin
// (a plain atomic per element serialises ~n same-address updates per pass).
__device__ __forceinline__ void hist_add_aggregated(int* hist, bool valid, uint32_t bin) {
  uint64_t todo = __ballot(valid);
  const int lane = threadIdx.x & 63;
  // up to 4 rounds of "combine everybody in the leader's bin"; what is left is spread over many bins
  // (low contention) and goes through plain LDS atomics.
  for (int round = 0; round < 4 && todo; ++round) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    // leader is wave-uniform (derived from a ballot): v_readlane_b32, not a ds_bpermute round trip through the LDS crossbar
    const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
    const uint64_t same = __ballot(valid && bin == lb) & todo;
    if (lane == leader) atomicAdd(&hist[lb], __popcll(same));
    todo &= ~same;
  }
  if ((todo >> lane) & 1) atomicAdd(&hist[bin], 1);
}

// k-th smallest key (0-based) of |v[0..n)| (keys = fp32 bit patterns, monotone for non-negative floats).
// KPT > 0: every thread keeps its KPT keys in REGISTERS, loaded once with all loads in flight together; the four radix
// passes and the successor count then never touch memory.  (Reading v[i] again in every pass made each of the ~36 loop
// iterations of each pass a dependent L2 round trip: 71 us per step at 96x96 for one workgroup's worth of arithmetic.)
// KPT == 0: generic path for n > 64 * blockDim.x (keys re-read from memory in every pass).
// Key 0xffffffff marks "no element" (never a valid |x| bit pattern below NaN; the latent is finite).
// THREE passes over digits of 11 + 11 + 10 bits (2048-bin histogram) instead of four over bytes.  |x0| after the clamp lives in a handful of
// binades: with 8-bit digits the first pass left half of all keys in one bin (sign + 7 exponent bits: everything in [0.5, 2) shares 0x3f); an
// 11-bit first digit (sign, exponent, two mantissa bits) leaves < 10 % of the keys valid for passes two and three, where a key costs one
// ballot.  Measured: the step's three kernels 76 -> 73 us at 96x96 - the FIRST pass (every key through the ballot-aggregated histogram add,
// 36 keys x 16 waves on one CU) is what the 60 us are, whatever the digit plan; the lever that is left is spreading that pass over more CUs.
constexpr int K22_RS_BINS = 2048;
// floor_key: keys below it are not candidates (the caller counted them: k is the rank among the keys >= floor_key); 0 = all keys.
template <int KPT>
__device__ __forceinline__ uint32_t radix_select(const float* v, const uint32_t* keys, int n, int k, int* hist, uint32_t* bc, uint32_t floor_key = 0u) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0, mask = 0;
#pragma unroll 1
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const int nb = pass == 2 ? 1024 : 2048;
    const uint32_t dmask = (uint32_t)(nb - 1);
    for (int i = tid; i < K22_RS_BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    if constexpr (KPT > 0) {
#pragma unroll
      for (int j = 0; j < KPT; ++j) {
        const uint32_t key = keys[j];
        const bool valid = key != 0xffffffffu && (key & mask) == prefix && key >= floor_key;
        hist_add_aggregated(hist, valid, (key >> shift) & dmask);
      }
    } else {
      for (int i0 = 0; i0 < n; i0 += blockDim.x) {
        const int i = i0 + tid;
        uint32_t key = 0;
        bool valid = false;
        if (i < n) {
          key = __float_as_uint(fabsf(v[i]));
          valid = (key & mask) == prefix;
        }
        hist_add_aggregated(hist, valid, (key >> shift) & dmask);
      }
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0: lane l owns bins [l * per, (l + 1) * per); inclusive scan of the lane totals, then the owner of rank k walks its bins
      const int per = nb >> 6;
      int run = 0;
      for (int j = 0; j < per; ++j) run += hist[tid * per + j];
      int incl = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (tid >= o) incl += t;
      }
      const int excl = incl - run;
      if (k >= excl && k < incl) {      // exactly one lane: the valid keys number more than k
        int acc = excl;
        for (int j = 0; j < per; ++j) {
          const int c = hist[tid * per + j];
          if (k < acc + c) { bc[0] = (uint32_t)(tid * per + j); bc[1] = (uint32_t)(k - acc); break; }
          acc += c;
        }
      }
    }
    __syncthreads();
    prefix |= bc[0] << shift;
    mask |= dmask << shift;
    k = (int)bc[1];
    __syncthreads();
  }
  return prefix;
}

template <int KPT>
__global__ __launch_bounds__(1024) void sampler_threshold_kernel(SamplerParams p) {
  __shared__ int hist[K22_RS_BINS];
  __shared__ uint32_t bc[2];
  __shared__ unsigned int succ_cnt[2];  // [0] = #keys <= a, [1] = min key > a
  const int n = 4 * p.HW;
  const int k_hi = p.n_lo + 1 < n ? p.n_lo + 1 : n - 1;
  uint32_t keys[KPT > 0 ? KPT : 1];
  if constexpr (KPT > 0) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int i = j * 1024 + (int)threadIdx.x;
      keys[j] = i < n ? __float_as_uint(fabsf(p.x0_buf[i])) : 0xffffffffu;
    }
  }
  // Round 6: the order statistic wanted is a HIGH percentile (99.5 %), and the first radix pass - every key through the ballot-aggregated
  // histogram add, 36 keys x 16 waves on one CU - was ~60 us of this kernel whatever the digit plan.  So the keys are filtered first, exactly:
  // a pivot is selected from a strided sample of 1024 keys (its 97.3rd percentile: five standard deviations of a sample quantile below
  // 99.5 %), the keys >= pivot are COUNTED (one compare + ballot per key), and if the wanted rank lies among them - it practically always
  // does - the three radix passes run on those ~3 % of the keys only, with the rank shifted by the number of keys below the pivot.  If it
  // does not (or n is small) the select runs over all keys as before: the result is the exact order statistic either way
  // (tests/test_kernels_gpu.py::test_sampler_threshold_is_the_exact_order_statistic_on_adversarial_keys).  Measured: the step's three
  // kernels 73 -> 59 us at 96x96 (tools/bench_sampler.py); what is left of this kernel is six short passes of fixed cost (bin clear,
  // barriers, the one-wave scan) and the key loads.
  uint32_t floor_key = 0u;
  int k_sel = p.n_lo;
  if constexpr (KPT >= 16) {
    __shared__ unsigned int n_ge;
    uint32_t skey = 0xffffffffu;
    const int js = (int)threadIdx.x % KPT;
#pragma unroll
    for (int j = 0; j < KPT; ++j) skey = (j == js) ? keys[j] : skey;    // element js * 1024 + tid: a stride through every channel / row
    unsigned int ns_local = skey != 0xffffffffu ? 1u : 0u;
    if (threadIdx.x == 0) n_ge = 0u;
    __syncthreads();
    if (ns_local) atomicAdd(&n_ge, 1u);     // number of valid samples (1024 unless n is not a multiple of 1024)
    __syncthreads();
    const int ns = (int)n_ge;
    __syncthreads();
    if (ns >= 512) {
      const uint32_t pivot = radix_select<1>(nullptr, &skey, ns, (int)(ns * 0.973f), hist, bc);
      if (threadIdx.x == 0) n_ge = 0u;
      __syncthreads();
      unsigned int c = 0;
#pragma unroll
      for (int j = 0; j < KPT; ++j) c += (keys[j] != 0xffffffffu && keys[j] >= pivot) ? 1u : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      if ((threadIdx.x & 63) == 0) atomicAdd(&n_ge, c);
      __syncthreads();
      const int below = n - (int)n_ge;      // keys < pivot
      if (p.n_lo >= below) { floor_key = pivot; k_sel = p.n_lo - below; }
      __syncthreads();
    }
  }
  const uint32_t ka = radix_select<KPT>(p.x0_buf, keys, n, k_sel, hist, bc, floor_key);
  // the next order statistic: a itself if enough keys are <= a, else the smallest key above a
  if (threadIdx.x == 0) { succ_cnt[0] = 0u; succ_cnt[1] = 0xffffffffu; }
  __syncthreads();
  unsigned int cnt = 0, mn = 0xffffffffu;
  if constexpr (KPT > 0) {
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const uint32_t key = keys[j];
      if (key == 0xffffffffu) continue;
      if (key <= ka) ++cnt;
      else mn = key < mn ? key : mn;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t key = __float_as_uint(fabsf(p.x0_buf[i]));
      if (key <= ka) ++cnt;
      else mn = key < mn ? key : mn;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    const unsigned int t = __shfl_xor(mn, o, 64);
    mn = t < mn ? t : mn;
  }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&succ_cnt[0], cnt); atomicMin(&succ_cnt[1], mn); }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t kb = ((int)succ_cnt[0] > k_hi || k_hi == p.n_lo) ? ka : succ_cnt[1];
    const float a = __uint_as_float(ka), b = __uint_as_float(kb);
    // numpy _lerp in float32: a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5
    const float t = (float)p.gamma;
    const float d = __fsub_rn(b, a);
    float r = __fadd_rn(a, __fmul_rn(d, t));
    if (t >= 0.5f) r = __fsub_rn(b, __fmul_rn(d, __fsub_rn(1.f, t)));
    p.s_buf[0] = r > 1.f ? r : 1.f;
  }
}

// pass 3: threshold, posterior mean, learned-range variance, ancestral noise.
__global__ __launch_bounds__(256) void sampler_final_kernel(SamplerParams p) {
  const float* tr = step_row(p);
  const float c1 = tr[2], c2 = tr[3], min_log = tr[4], max_log = tr[5], nonzero = tr[6];
  const float s = (p.n_lo >= 0) ? p.s_buf[0] : 0.f;
  const int64_t total = (int64_t)p.N * 4 * p.HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % p.HW);
    const int c = (int)((i / p.HW) % 4), n = (int)(i / (4 * (int64_t)p.HW));
    float x0 = p.x0_buf[i];
    if (p.n_lo >= 0) x0 = fminf(fmaxf(x0, -s), s) / s;
    const float mean = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, p.x[i]));
    const float v = p.model_out[((int64_t)n * 8 + 4 + c) * p.HW + pix];
    const float frac = (v + 1.f) / 2.f;
    const float logvar = __fadd_rn(__fmul_rn(frac, max_log), __fmul_rn(1.f - frac, min_log));
    p.x_out[i] = mean + nonzero * expf(0.5f * logvar) * p.noise[i];
    if (p.x0_out != nullptr) p.x0_out[i] = x0;
  }
}

// ---- DDIM step (kandinsky2/model/samplers.py:290-331, eta-general) with model_fn's classifier-free guidance folded in ----
// tab = (alpha_t, alpha_prev, sigma_t, sqrt(1 - alpha_t)) as fp32 (the reference builds them with torch.full from the fp64
// numpy schedule, i.e. rounded to fp32); e_t = u + g (c - u) on channels 0-3 of the raw UNet output.
__global__ __launch_bounds__(256) void ddim_step_kernel(const float* x, const float* model_out, const float* noise, const float* tab,
                                                        float guidance, int use_cfg, float* x_out, float* x0_out, int N, int HW) {
  const float a_t = tab[0], a_prev = tab[1], sigma = tab[2], s1m = tab[3];
  const float sq_at = sqrtf(a_t), sq_ap = sqrtf(a_prev);
  const float dirc = sqrtf(__fsub_rn(__fsub_rn(1.0f, a_prev), __fmul_rn(sigma, sigma)));
  const int64_t total = (int64_t)N * 4 * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % 4), n = (int)(i / (4 * (int64_t)HW));
    float e;
    if (use_cfg) {
      const int bs = N / 2, j = n % bs;
      const float ce = model_out[((int64_t)j * 8 + c) * HW + pix], ue = model_out[((int64_t)(j + bs) * 8 + c) * HW + pix];
      e = ue + guidance * (ce - ue);
    } else {
      e = model_out[((int64_t)n * 8 + c) * HW + pix];
    }
    const float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(s1m, e)), sq_at);
    float xp = __fadd_rn(__fmul_rn(sq_ap, x0), __fmul_rn(dirc, e));
    if (noise != nullptr) xp = __fadd_rn(xp, __fmul_rn(sigma, noise[i]));
    x_out[i] = xp;
    if (x0_out != nullptr) x0_out[i] = x0;
  }
}

// ---- PLMS step (kandinsky2/model/samplers.py:566-637, eta = 0 enforced at :355) with model_fn's guidance folded in --------
// e_t = u + g (c - u) of THIS model call; e' by order (history h1 = newest):
//   0: e_t   (first stage of the pseudo improved Euler start)      4: (h1 + e_t) / 2   (its second stage: h1 = e_t of stage one)
//   1: (3 e_t - h1) / 2      2: (23 e_t - 16 h1 + 5 h2) / 12      3: (55 e_t - 59 h1 + 37 h2 - 9 h3) / 24
// evaluated left to right with separate roundings like the reference's tensor expression; then the eta = 0 DDIM update
// x_out = sqrt(a_prev) (x - sqrt(1-a_t) e') / sqrt(a_t) + sqrt(1 - a_prev) e'.  e_store (optional) receives e_t.
__global__ __launch_bounds__(256) void plms_step_kernel(const float* x, const float* model_out, const float* h1, const float* h2, const float* h3,
                                                        int order, const float* tab, float guidance, int use_cfg, float* x_out, float* e_store,
                                                        float* x0_out, int N, int HW) {
  const float a_t = tab[0], a_prev = tab[1], s1m = tab[3];
  const float sq_at = sqrtf(a_t), sq_ap = sqrtf(a_prev);
  const float dirc = sqrtf(__fsub_rn(__fsub_rn(1.0f, a_prev), 0.0f));
  const int64_t total = (int64_t)N * 4 * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % 4), n = (int)(i / (4 * (int64_t)HW));
    float e;
    if (use_cfg) {
      const int bs = N / 2, j = n % bs;
      const float ce = model_out[((int64_t)j * 8 + c) * HW + pix], ue = model_out[((int64_t)(j + bs) * 8 + c) * HW + pix];
      e = __fadd_rn(ue, __fmul_rn(guidance, __fsub_rn(ce, ue)));
    } else {
      e = model_out[((int64_t)n * 8 + c) * HW + pix];
    }
    float ep = e;
    if (order == 4) ep = __fdiv_rn(__fadd_rn(h1[i], e), 2.0f);
    else if (order == 1) ep = __fdiv_rn(__fsub_rn(__fmul_rn(3.0f, e), h1[i]), 2.0f);
    else if (order == 2) ep = __fdiv_rn(__fadd_rn(__fsub_rn(__fmul_rn(23.0f, e), __fmul_rn(16.0f, h1[i])), __fmul_rn(5.0f, h2[i])), 12.0f);
    else if (order == 3)
      ep = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(55.0f, e), __fmul_rn(59.0f, h1[i])), __fmul_rn(37.0f, h2[i])), __fmul_rn(9.0f, h3[i])), 24.0f);
    const float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(s1m, ep)), sq_at);
    x_out[i] = __fadd_rn(__fmul_rn(sq_ap, x0), __fmul_rn(dirc, ep));
    if (e_store != nullptr) e_store[i] = e;
    if (x0_out != nullptr) x0_out[i] = x0;
  }
}

int launch_plms_step(const float* x, const float* model_out, const float* h1, const float* h2, const float* h3, int order, const float* tab,
                     float guidance, int use_cfg, float* x_out, float* e_store, float* x0_out, int N, int HW, hipStream_t s) {
  if (N <= 0 || HW <= 0 || (use_cfg && (N & 1))) return k22_set_error(K22_EINVAL, "plms_step: bad batch");
  if (order < 0 || order > 4 || ((order == 1 || order == 4) && !h1) || (order == 2 && (!h1 || !h2)) || (order == 3 && (!h1 || !h2 || !h3)))
    return k22_set_error(K22_EINVAL, "plms_step: order 0-4 with the eps history it needs");
  const int64_t total = (int64_t)N * 4 * HW;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(plms_step_kernel, dim3(nb), dim3(256), 0, s, x, model_out, h1, h2, h3, order, tab, guidance, use_cfg, x_out, e_store, x0_out, N, HW);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

// ---- inpainting mask pre-step (prepare_mask, kandinsky2/utils.py:11-31) ----------------------------------------------------
// The reference walks the latent-resolution mask in a Python double loop and, for every pixel whose ORIGINAL value is not 1,
// zeroes six neighbours: up, left, up-left, down, right, down-right (not the anti-diagonal ones).  As a gather: a pixel
// becomes 0 when any of the six in-bounds pixels it is such a neighbour OF had an original value != 1; else it keeps its
// value.  mask: fp32 [C][H][W] (channel 0 decides, all channels are written, as mask[:, i, j] = 0 does).
__global__ __launch_bounds__(256) void prepare_mask_kernel(const float* old_mask, float* out, int C, int H, int W) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  // target (y, x) = source + d  for d in {(-1,0), (0,-1), (-1,-1), (1,0), (0,1), (1,1)}  =>  source = target - d
  const int sy[6] = {y + 1, y, y + 1, y - 1, y, y - 1};
  const int sx[6] = {x, x + 1, x + 1, x, x - 1, x - 1};
  bool kill = false;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (sy[k] >= 0 && sy[k] < H && sx[k] >= 0 && sx[k] < W && old_mask[sy[k] * W + sx[k]] != 1.0f) kill = true;
  for (int c = 0; c < C; ++c) out[(c * H + y) * W + x] = kill ? 0.0f : old_mask[(c * H + y) * W + x];
}

int launch_prepare_mask(const float* old_mask, float* out, int C, int H, int W, hipStream_t s) {
  if (C < 1 || H < 1 || W < 1 || old_mask == out) return k22_set_error(K22_EINVAL, "prepare_mask: bad arguments (out of place only)");
  hipLaunchKernelGGL(prepare_mask_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, old_mask, out, C, H, W);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

int launch_ddim_step(const float* x, const float* model_out, const float* noise, const float* tab, float guidance, int use_cfg,
                     float* x_out, float* x0_out, int N, int HW, hipStream_t s) {
  if (N <= 0 || HW <= 0 || (use_cfg && (N & 1))) return k22_set_error(K22_EINVAL, "ddim_step: bad batch");
  const int64_t total = (int64_t)N * 4 * HW;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(ddim_step_kernel, dim3(nb), dim3(256), 0, s, x, model_out, noise, tab, guidance, use_cfg, x_out, x0_out, N, HW);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

// advances the device step counter (one thread); used when a whole step is replayed as a graph.
__global__ void step_advance_kernel(int* step, int delta) { *step += delta; }

int launch_sampler_step(const SamplerParams& p, hipStream_t s) {
  if (p.N <= 0 || p.HW <= 0) return k22_set_error(K22_EINVAL, "sampler: empty batch");
  if (p.use_cfg && (p.N & 1)) return k22_set_error(K22_EINVAL, "sampler: CFG needs an even batch [cond | uncond]");
  if ((p.mask == nullptr) != (p.init_img == nullptr)) return k22_set_error(K22_EINVAL, "sampler: init_img and mask go together");
  const int64_t total = (int64_t)p.N * 4 * p.HW;
  int nb = (int)((total + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(sampler_x0_kernel, dim3(nb), dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  if (p.n_lo >= 0) {
    const int kpt = (4 * p.HW + 1023) / 1024;   // keys per thread of the one 1024-thread workgroup
    if (kpt <= 4) hipLaunchKernelGGL(sampler_threshold_kernel<4>, dim3(1), dim3(1024), 0, s, p);
    else if (kpt <= 16) hipLaunchKernelGGL(sampler_threshold_kernel<16>, dim3(1), dim3(1024), 0, s, p);
    else if (kpt <= 36) hipLaunchKernelGGL(sampler_threshold_kernel<36>, dim3(1), dim3(1024), 0, s, p);
    else if (kpt <= 64) hipLaunchKernelGGL(sampler_threshold_kernel<64>, dim3(1), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL(sampler_threshold_kernel<0>, dim3(1), dim3(1024), 0, s, p);
    K22_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(sampler_final_kernel, dim3(nb), dim3(256), 0, s, p);
  K22_CHECK_LAUNCH();
  return K22_OK;
}

int launch_step_advance(int* step, int delta, hipStream_t s) {
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, step, delta);
  K22_CHECK_LAUNCH();
  return K22_OK;
}
