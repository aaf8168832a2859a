// Conservation fixers of the reference PostBlock (credit/postblock/gen1.py) on the device:
//   TracerFixer :136-167, GlobalMassFixer :280-391, GlobalWaterFixer :489-569, GlobalEnergyFixer :704-822,
// on pressure levels (credit/physics_core.py::physics_pressure_level :75-297) and on hybrid sigma-pressure levels
// (physics_hybrid_sigma_level :300-520: p = a_l + b_l * surface pressure; the mass fixer then rescales the surface
// pressure instead of q, gen1.py:355-375).  SURVEY.md §8(a) a12.
//
// Every global fixer is "column integrals -> a few area-weighted global sums -> one scalar ratio -> elementwise
// correction".  HBM-bound integer-free float work: one thread per grid cell walks its column (loads are coalesced
// along longitude for every level), per-cell integrals in fp32 like the reference, the global sums in fp64 with a
// fixed two-stage order (per-workgroup partials, then one workgroup) so results are bit-reproducible, then an
// elementwise apply kernel.  Tensors are the caller's fp32 NCHW buffers (y [C_out][H][W], x [C_in][frames][H][W]).
#pragma once
#include <vector>

#include "wx_common.h"

namespace wx {

constexpr float kGravity = 9.80665f, kRhoWater = 1000.0f, kLhWater = 2.501e6f, kCpDry = 1004.64f, kCpVapor = 1810.0f;
constexpr double kRadEarth = 6371000.0;
constexpr int kMaxLevels = 64;

struct FixParams {
  const float* x;        // input  [c_in][frames][HW]; the LAST frame is used
  float* y;              // output [c_out][HW], fixed in place
  int hw, c_in, frames, c_out;
  const float* area;     // [HW]
  const float* p;        // [n_p] pressure levels (Pa)            (pressure grids)
  const float *ca, *cb;  // [n_p] hybrid coefficients a (Pa), b    (sigma grids)
  int sigma, sp_ind;     // sigma grid flag; surface-pressure channel (same index in x and y, gen1.py:306-308)
  int n_p, midpoint;
  const float *mean_in, *std_in, *mean_out, *std_out;  // nullptr unless denorm
  // op
  int kind;              // 1 mass, 2 water, 3 energy
  int q0, nlev;          // q block start / levels carried per 3-D variable
  int ind_fix, ind_fix_start;
  int precip, evapor;
  int T0, U0, V0;
  // energy fixers: R_T = sum_k toa_s[k] * y[toa_i[k]], F_S = sum_k srf_s[k] * y[srf_i[k]] (left to right, as the reference
  // writes them: GlobalEnergyFixer gen1.py:762-768 all +; GlobalEnergyFixerUpDown :982-994 with the up/down signs)
  int toa_n, srf_n, toa_i[4], srf_i[8];
  float toa_s[4], srf_s[8];
  const float* gph;      // [HW]
  float n_seconds;
  double* partial;       // [blocks][4]
  double* sums;          // [4]
  float* ratio;          // [1]
  int n_blocks;
};

__device__ inline float fx_in(const FixParams& p, int ch, int cell) {
  const float v = p.x[((int64_t)ch * p.frames + (p.frames - 1)) * p.hw + cell];
  return p.mean_in ? v * p.std_in[ch] + p.mean_in[ch] : v;
}
__device__ inline float fx_out(const FixParams& p, int ch, int cell) {
  const float v = p.y[(int64_t)ch * p.hw + cell];
  return p.mean_out ? v * p.std_out[ch] + p.mean_out[ch] : v;
}
// pressure of level l in this column: the level table, or a_l + b_l * sp on hybrid sigma grids (physics_core.py:386)
__device__ inline float lev_p(const FixParams& p, int l, float sp) { return p.sigma ? p.ca[l] + p.cb[l] * sp : p.p[l]; }
// pressure integral of f(l) over levels [a, b): trapz over p[a..b-1], or midpoint with thickness diff(p)
template <typename F>
__device__ inline float col_integral(const FixParams& p, int a, int b, F f, float sp = 0.f) {
  float acc = 0.f;
  if (p.midpoint) {
    for (int l = a; l < b; ++l) acc += f(l) * (lev_p(p, l + 1, sp) - lev_p(p, l, sp));
  } else {
    float prev = f(a);
    for (int l = a; l + 1 < b; ++l) {
      const float cur = f(l + 1);
      acc += 0.5f * (prev + cur) * (lev_p(p, l + 1, sp) - lev_p(p, l, sp));
      prev = cur;
    }
  }
  return acc;
}

__global__ __launch_bounds__(256) void fix_reduce_kernel(const FixParams p) {
  __shared__ double sh[4][256];
  const int cell = blockIdx.x * 256 + threadIdx.x;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (cell < p.hw) {
    const double a = (double)p.area[cell];
    const int nl = p.nlev;
    const float sp_in = p.sigma ? fx_in(p, p.sp_ind, cell) : 0.f, sp_pr = p.sigma ? fx_out(p, p.sp_ind, cell) : 0.f;
    if (p.kind == 1 && p.sigma) {
      // dry mass at t0 with the input surface pressure; at t1 split into the a- and b-parts of dp (gen1.py:357-371)
      const float i0 = col_integral(p, 0, nl, [&](int l) { return 1.f - fx_in(p, p.q0 + l, cell); }, sp_in) / kGravity;
      float pa = 0.f, pb = 0.f;
      if (p.midpoint) {
        for (int l = 0; l < nl; ++l) {
          const float dry = 1.f - fx_out(p, p.q0 + l, cell);
          pa += (p.ca[l + 1] - p.ca[l]) * dry;
          pb += (p.cb[l + 1] - p.cb[l]) * dry;
        }
      } else {
        for (int l = 0; l + 1 < nl; ++l) {
          const float dry = 1.f - (fx_out(p, p.q0 + l, cell) + fx_out(p, p.q0 + l + 1, cell)) / 2.f;
          pa += (p.ca[l + 1] - p.ca[l]) * dry;
          pb += (p.cb[l + 1] - p.cb[l]) * dry;
        }
      }
      s[0] = i0 * a; s[1] = (double)pa * a / (double)kGravity; s[2] = (double)(pb * sp_pr) * a / (double)kGravity;
    } else if (p.kind == 1) {
      const float i0 = col_integral(p, 0, nl, [&](int l) { return 1.f - fx_in(p, p.q0 + l, cell); }) / kGravity;
      const float ih = col_integral(p, 0, p.ind_fix, [&](int l) { return 1.f - fx_out(p, p.q0 + l, cell); }) / kGravity;
      const float ifx = col_integral(p, p.ind_fix_start, nl, [&](int l) { return 1.f - fx_out(p, p.q0 + l, cell); }) / kGravity;
      s[0] = i0 * a; s[1] = ih * a; s[2] = ifx * a;
    } else if (p.kind == 2) {
      const float t_in = col_integral(p, 0, nl, [&](int l) { return fx_in(p, p.q0 + l, cell); }, sp_in) / kGravity;
      const float t_pr = col_integral(p, 0, nl, [&](int l) { return fx_out(p, p.q0 + l, cell); }, sp_pr) / kGravity;
      s[0] = (double)((t_pr - t_in) / p.n_seconds) * a;
      s[1] = (double)(fx_out(p, p.evapor, cell) * kRhoWater / p.n_seconds) * a;
      s[2] = (double)(fx_out(p, p.precip, cell) * kRhoWater / p.n_seconds) * a;
    } else {
      const float gph = p.gph[cell];
      float rt = p.toa_s[0] * fx_out(p, p.toa_i[0], cell), fs = p.srf_s[0] * fx_out(p, p.srf_i[0], cell);
      for (int k = 1; k < p.toa_n; ++k) rt += p.toa_s[k] * fx_out(p, p.toa_i[k], cell);
      for (int k = 1; k < p.srf_n; ++k) fs += p.srf_s[k] * fx_out(p, p.srf_i[k], cell);
      rt /= p.n_seconds;
      fs /= p.n_seconds;
      const float te0 = col_integral(p, 0, nl, [&](int l) {
        const float q = fx_in(p, p.q0 + l, cell), u = fx_in(p, p.U0 + l, cell), v = fx_in(p, p.V0 + l, cell);
        const float cp = (1.f - q) * kCpDry + q * kCpVapor;
        return cp * fx_in(p, p.T0 + l, cell) + (kLhWater * q + gph + 0.5f * (u * u + v * v));
      }, sp_in) / kGravity;
      const float te1 = col_integral(p, 0, nl, [&](int l) {
        const float q = fx_out(p, p.q0 + l, cell), u = fx_out(p, p.U0 + l, cell), v = fx_out(p, p.V0 + l, cell);
        const float cp = (1.f - q) * kCpDry + q * kCpVapor;
        return cp * fx_out(p, p.T0 + l, cell) + (kLhWater * q + gph + 0.5f * (u * u + v * v));
      }, sp_pr) / kGravity;
      s[0] = (double)rt * a; s[1] = (double)fs * a; s[2] = (double)te0 * a; s[3] = (double)te1 * a;
    }
  }
  for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (unsigned o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) p.partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = sh[threadIdx.x][0];
}

// block partials -> p.sums[4] (fixed order: deterministic)
__global__ __launch_bounds__(256) void fix_sum_kernel(const FixParams p) {
  __shared__ double sh[4][256];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < p.n_blocks; b += 256)
    for (int k = 0; k < 4; ++k) s[k] += p.partial[(int64_t)b * 4 + k];
  for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  for (unsigned o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) p.sums[threadIdx.x] = sh[threadIdx.x][0];
}
// p.sums[4] (of the whole globe: under lat-band sharding the engine has added every rank's sums by now) -> correction ratio
__global__ void fix_ratio_kernel(const FixParams p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double a = p.sums[0], b = p.sums[1], c = p.sums[2], d = p.sums[3];
  double r;
  if (p.kind == 1) r = (a - b) / c;                                   // (M_dry(t0) - M_hold(t1)) / M_fix(t1)
  else if (p.kind == 2) r = (c + (-a - b - c)) / c;                    // (P + residual) / P
  else r = ((double)p.n_seconds * (a - b) + c) / d;                   // (dt (R_T - F_S) + TE(t0)) / TE(t1)
  p.ratio[0] = (float)r;
}

__global__ __launch_bounds__(256) void fix_apply_kernel(const FixParams p) {
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= p.hw) return;
  const float r = p.ratio[0];
  auto put = [&](int ch, float v) {
    if (p.mean_out) v = (v - p.mean_out[ch]) / p.std_out[ch];
    p.y[(int64_t)ch * p.hw + cell] = v;
  };
  if (p.kind == 1 && p.sigma) {
    put(p.sp_ind, fx_out(p, p.sp_ind, cell) * r);   // gen1.py:373-375: sp_pred * sp_correct_ratio
  } else if (p.kind == 1) {
    for (int l = p.ind_fix_start; l < p.nlev; ++l) put(p.q0 + l, 1.f - (1.f - fx_out(p, p.q0 + l, cell)) * r);
  } else if (p.kind == 2) {
    put(p.precip, fx_out(p, p.precip, cell) * r);
  } else {
    const float gph = p.gph[cell];
    for (int l = 0; l < p.nlev; ++l) {
      const float q = fx_out(p, p.q0 + l, cell), u = fx_out(p, p.U0 + l, cell), v = fx_out(p, p.V0 + l, cell);
      const float cp = (1.f - q) * kCpDry + q * kCpVapor;
      const float eq = kLhWater * q + gph + 0.5f * (u * u + v * v);
      const float e1 = cp * fx_out(p, p.T0 + l, cell) + eq;
      put(p.T0 + l, (e1 * r - eq) / cp);
    }
  }
}

// TracerFixer on the fp32 NCHW tensor (the engine's tail kernel has its own fused copy)
__global__ __launch_bounds__(256) void fix_tracer_kernel(float* y, int hw, int n, const int* inds, const float* lo, const float* hi,
                                                         const float* mean, const float* stdv) {
  const int cell = blockIdx.x * 256 + threadIdx.x;
  if (cell >= hw) return;
  for (int k = 0; k < n; ++k) {
    const int ch = inds[k];
    float v = y[(int64_t)ch * hw + cell];
    if (mean) v = v * stdv[ch] + mean[ch];
    v = v < lo[k] ? lo[k] : v;
    v = v >= hi[k] ? hi[k] : v;
    if (mean) v = (v - mean[ch]) / stdv[ch];
    y[(int64_t)ch * hw + cell] = v;
  }
}

// ----------------------------------------------------------------------------------------------------------------
struct PostOp {
  int kind = 0;  // 0 tracer, 1 mass, 2 water, 3 energy
  int denorm = 0;
  int q0 = 0, nlev = 0, fix_level_num = 0, precip = 0, evapor = 0, T0 = 0, U0 = 0, V0 = 0;
  int toa_n = 0, srf_n = 0, toa_i[4] = {0, 0, 0, 0}, srf_i[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float toa_s[4] = {0, 0, 0, 0}, srf_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float n_seconds = 0.f;
  float* gph = nullptr;
  int n_tr = 0;
  int* tr_inds = nullptr;
  float *tr_lo = nullptr, *tr_hi = nullptr;
};

class PostBlock {
 public:
  PostBlock(int H, int W, int c_in, int frames, int c_out, int dev)
      : h(H), w(W), cin(c_in), fr(frames), cout(c_out), device(dev), h_full(H) {
    if (H < 3 || W < 3 || c_in < 1 || c_out < 1 || frames < 1) throw std::runtime_error("wx_post_create: bad geometry");
    WX_HIP(hipSetDevice(device));
    n_blocks = cdiv((int64_t)H * W, 256);
    partial = (double*)alloc((size_t)n_blocks * 4 * sizeof(double));
    sums = (double*)alloc(4 * sizeof(double));
    ratio = (float*)alloc(sizeof(float));
  }
  ~PostBlock() {
    (void)hipSetDevice(device);
    for (void* p : allocs) (void)hipFree(p);
  }
  int h, w, cin, fr, cout, device, n_blocks = 0;
  int h_full, row0 = 0;   // lat-band mode: the block covers rows [row0, row0 + h) of a grid of h_full rows
  std::vector<void*> allocs;
  float *area = nullptr, *plev = nullptr, *coef_a = nullptr, *coef_b = nullptr;
  int n_p = 0, midpoint = 0, sigma = 0, sp_ind = -1;
  float *mean_in = nullptr, *std_in = nullptr, *mean_out = nullptr, *std_out = nullptr;
  double *partial = nullptr, *sums = nullptr;
  float* ratio = nullptr;
  std::vector<PostOp> ops;
  std::vector<double> last_sums;

  void* alloc(size_t bytes) {
    void* p = nullptr;
    WX_HIP(hipMalloc(&p, bytes));
    allocs.push_back(p);
    return p;
  }
  float* upload(const float* src, size_t n) {
    float* d = (float*)alloc(n * sizeof(float));
    WX_HIP(hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice));
    return d;
  }
  // Lat-band mode (wx_band.h): this block then sees only rows [r0, r0 + rows) of x and y; the grid arrays passed later are
  // still those of the WHOLE grid (cell areas need the neighbouring latitudes), and the fixers' global sums are completed
  // by the engine between reduce_op and finish_op.
  void set_band(int r0, int rows) {
    if (area || !ops.empty()) throw std::runtime_error("wx_post_set_band: call it before the grid and the fixers are set");
    if (r0 < 0 || rows < 0 || r0 + rows > h_full) throw std::runtime_error("wx_post_set_band: rows outside the grid");
    row0 = r0; h = rows;
    n_blocks = (int)cdiv((int64_t)h * w, 256);
  }
  // |R^2 d(sin lat) d(lon)|, second-order one-sided differences at the edges, lon difference wrapped into (-pi, pi]
  // (credit/physics_core.py:113-125: torch.gradient(edge_order=2)); fp32 like the reference
  void set_grid(const float* lat2d, const float* lon2d, const float* p_levels, int n_levels, int mid) {
    if (n_levels < 2 || n_levels > kMaxLevels) throw std::runtime_error("wx_post_set_grid: 2..64 pressure levels");
    WX_HIP(hipSetDevice(device));
    set_area(lat2d, lon2d);
    plev = upload(p_levels, n_levels);
    n_p = n_levels;
    midpoint = mid;
    sigma = 0;
  }
  // hybrid sigma-pressure grid: physics_hybrid_sigma_level(lon2d, lat2d, coef_a, coef_b, midpoint) (physics_core.py:314-368)
  void set_grid_sigma(const float* lat2d, const float* lon2d, const float* ca, const float* cb, int n_levels, int mid, int sp) {
    if (n_levels < 2 || n_levels > kMaxLevels) throw std::runtime_error("wx_post_set_grid_sigma: 2..64 levels");
    if (sp < 0 || sp >= cout || sp >= cin) throw std::runtime_error("wx_post_set_grid_sigma: surface-pressure channel out of range");
    WX_HIP(hipSetDevice(device));
    set_area(lat2d, lon2d);
    coef_a = upload(ca, n_levels);
    coef_b = upload(cb, n_levels);
    n_p = n_levels;
    midpoint = mid;
    sigma = 1;
    sp_ind = sp;
  }
  void set_area(const float* lat2d, const float* lon2d) {
    const int H = h_full;
    std::vector<float> a((size_t)H * w);
    const float d2r = 3.14159265358979323846f / 180.f;
    auto sl = [&](int i, int j) { return std::sin(lat2d[(size_t)i * w + j] * d2r); };
    auto lo = [&](int i, int j) { return lon2d[(size_t)i * w + j] * d2r; };
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < w; ++j) {
        float dphi, dlam;
        if (i == 0) dphi = (-3.f * sl(0, j) + 4.f * sl(1, j) - sl(2, j)) / 2.f;
        else if (i == H - 1) dphi = (3.f * sl(H - 1, j) - 4.f * sl(H - 2, j) + sl(H - 3, j)) / 2.f;
        else dphi = (sl(i + 1, j) - sl(i - 1, j)) / 2.f;
        if (j == 0) dlam = (-3.f * lo(i, 0) + 4.f * lo(i, 1) - lo(i, 2)) / 2.f;
        else if (j == w - 1) dlam = (3.f * lo(i, w - 1) - 4.f * lo(i, w - 2) + lo(i, w - 3)) / 2.f;
        else dlam = (lo(i, j + 1) - lo(i, j - 1)) / 2.f;
        const float pi = 3.14159265358979323846f;
        float t = std::fmod(dlam + pi, 2.f * pi);       // python %: result takes the sign of the divisor
        if (t < 0.f) t += 2.f * pi;
        dlam = t - pi;
        a[(size_t)i * w + j] = std::fabs((float)(kRadEarth * kRadEarth) * dphi * dlam);
      }
    area = upload(a.data() + (size_t)row0 * w, std::max<size_t>((size_t)h * w, 1));
  }
  void set_stats(const float* mi, const float* si, const float* mo, const float* so) {
    WX_HIP(hipSetDevice(device));
    mean_in = upload(mi, cin); std_in = upload(si, cin);
    mean_out = upload(mo, cout); std_out = upload(so, cout);
  }
  void need_grid() const { if (!area) throw std::runtime_error("wx_post: call wx_post_set_grid first"); }
  void need_stats(int denorm) const { if (denorm && !mean_out) throw std::runtime_error("wx_post: denorm needs wx_post_set_stats first"); }
  void check_block(int start, int n, int limit, const char* what) const {
    if (start < 0 || n < 1 || start + n > limit) throw std::runtime_error(std::string("wx_post: channel block out of range: ") + what);
  }
  int levels_carried() const { return midpoint ? n_p - 1 : n_p; }

  void add_tracer(const int32_t* inds, const float* thres, const float* thres_max, int n, int denorm) {
    need_stats(denorm);
    WX_HIP(hipSetDevice(device));
    PostOp op;
    op.kind = 0; op.denorm = denorm; op.n_tr = n;
    std::vector<float> hi(n, 3.4e38f);
    for (int i = 0; i < n; ++i) {
      if (inds[i] < 0 || inds[i] >= cout) throw std::runtime_error("wx_post: tracer index out of range");
      if (thres_max) hi[i] = thres_max[i];
    }
    op.tr_inds = (int*)alloc(n * sizeof(int));
    WX_HIP(hipMemcpy(op.tr_inds, inds, n * sizeof(int), hipMemcpyHostToDevice));
    op.tr_lo = upload(thres, n);
    op.tr_hi = upload(hi.data(), n);
    ops.push_back(op);
  }
  void add_mass(int q0, int fix_level_num, int denorm) {
    need_grid(); need_stats(denorm);
    PostOp op;
    op.kind = 1; op.denorm = denorm; op.q0 = q0; op.nlev = levels_carried(); op.fix_level_num = fix_level_num;
    check_block(q0, op.nlev, cout, "q (output)"); check_block(q0, op.nlev, cin, "q (input)");
    if (fix_level_num < 1 || fix_level_num > n_p) throw std::runtime_error("wx_post: fix_level_num out of range");
    ops.push_back(op);
  }
  void add_water(int q0, int precip, int evapor, float n_seconds, int denorm) {
    need_grid(); need_stats(denorm);
    PostOp op;
    op.kind = 2; op.denorm = denorm; op.q0 = q0; op.nlev = levels_carried(); op.precip = precip; op.evapor = evapor;
    op.n_seconds = n_seconds;
    check_block(q0, op.nlev, cout, "q (output)"); check_block(q0, op.nlev, cin, "q (input)");
    check_block(precip, 1, cout, "precip"); check_block(evapor, 1, cout, "evapor");
    ops.push_back(op);
  }
  void add_energy(int T0, int q0, int U0, int V0, const int32_t rad[6], const float* gph_surf, float n_seconds, int denorm) {
    need_grid(); need_stats(denorm);
    WX_HIP(hipSetDevice(device));
    PostOp op;
    op.kind = 3; op.denorm = denorm; op.T0 = T0; op.q0 = q0; op.U0 = U0; op.V0 = V0; op.nlev = levels_carried();
    op.toa_n = 2; op.srf_n = 4;
    for (int k = 0; k < 2; ++k) { op.toa_i[k] = rad[k]; op.toa_s[k] = 1.f; }
    for (int k = 0; k < 4; ++k) { op.srf_i[k] = rad[2 + k]; op.srf_s[k] = 1.f; }
    finish_energy(op, T0, q0, U0, V0, gph_surf, n_seconds);
  }
  // GlobalEnergyFixerUpDown (gen1.py:825-1030): flux = [TOA down solar, TOA up solar, TOA up OLR, surf down solar, surf up solar,
  // surf down LW, surf up LW, SH, LH];  R_T = d - u - olr,  F_S = ds - us + dl - ul - sh - lh
  void add_energy_updown(int T0, int q0, int U0, int V0, const int32_t flux[9], const float* gph_surf, float n_seconds, int denorm) {
    need_grid(); need_stats(denorm);
    WX_HIP(hipSetDevice(device));
    PostOp op;
    op.kind = 3; op.denorm = denorm; op.T0 = T0; op.q0 = q0; op.U0 = U0; op.V0 = V0; op.nlev = levels_carried();
    const float ts[3] = {1.f, -1.f, -1.f}, ss[6] = {1.f, -1.f, 1.f, -1.f, -1.f, -1.f};
    op.toa_n = 3; op.srf_n = 6;
    for (int k = 0; k < 3; ++k) { op.toa_i[k] = flux[k]; op.toa_s[k] = ts[k]; }
    for (int k = 0; k < 6; ++k) { op.srf_i[k] = flux[3 + k]; op.srf_s[k] = ss[k]; }
    finish_energy(op, T0, q0, U0, V0, gph_surf, n_seconds);
  }
  // generic form: R_T = sum_k toa_sign[k] * y[toa_ind[k]], F_S = sum_k srf_sign[k] * y[srf_ind[k]] (the gen-2 fixer,
  // credit/postblock/conservation.py:344-358, uses yet another sign convention than the two gen-1 classes)
  void add_energy_signed(int T0, int q0, int U0, int V0, int n_toa, const int32_t* toa_i, const float* toa_s, int n_srf,
                         const int32_t* srf_i, const float* srf_s, const float* gph_surf, float n_seconds, int denorm) {
    need_grid(); need_stats(denorm);
    if (n_toa < 1 || n_toa > 4 || n_srf < 1 || n_srf > 8) throw std::runtime_error("wx_post: 1..4 TOA and 1..8 surface flux terms");
    WX_HIP(hipSetDevice(device));
    PostOp op;
    op.kind = 3; op.denorm = denorm; op.T0 = T0; op.q0 = q0; op.U0 = U0; op.V0 = V0; op.nlev = levels_carried();
    op.toa_n = n_toa; op.srf_n = n_srf;
    for (int k = 0; k < n_toa; ++k) { op.toa_i[k] = toa_i[k]; op.toa_s[k] = toa_s[k]; }
    for (int k = 0; k < n_srf; ++k) { op.srf_i[k] = srf_i[k]; op.srf_s[k] = srf_s[k]; }
    finish_energy(op, T0, q0, U0, V0, gph_surf, n_seconds);
  }
  void finish_energy(PostOp& op, int T0, int q0, int U0, int V0, const float* gph_surf, float n_seconds) {
    op.n_seconds = n_seconds;
    for (int s : {T0, q0, U0, V0}) { check_block(s, op.nlev, cout, "3-D block (output)"); check_block(s, op.nlev, cin, "3-D block (input)"); }
    for (int k = 0; k < op.toa_n; ++k) check_block(op.toa_i[k], 1, cout, "flux channel");
    for (int k = 0; k < op.srf_n; ++k) check_block(op.srf_i[k], 1, cout, "flux channel");
    op.gph = upload(gph_surf + (size_t)row0 * w, std::max<size_t>((size_t)h * w, 1));
    ops.push_back(op);
  }

  FixParams fix_params(const PostOp& op, const float* x, float* y) const {
    FixParams p;
    std::memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.hw = h * w; p.c_in = cin; p.frames = fr; p.c_out = cout;
    p.area = area; p.p = plev; p.ca = coef_a; p.cb = coef_b; p.sigma = sigma; p.sp_ind = sp_ind; p.n_p = n_p; p.midpoint = midpoint;
    if (op.denorm) { p.mean_in = mean_in; p.std_in = std_in; p.mean_out = mean_out; p.std_out = std_out; }
    p.kind = op.kind; p.q0 = op.q0; p.nlev = op.nlev;
    p.ind_fix = n_p - op.fix_level_num + 1;                 // gen1.py:224 / :264
    p.ind_fix_start = midpoint ? p.ind_fix : p.ind_fix - 1;  // gen1.py:267-270
    p.precip = op.precip; p.evapor = op.evapor;
    p.T0 = op.T0; p.U0 = op.U0; p.V0 = op.V0;
    p.toa_n = op.toa_n; p.srf_n = op.srf_n;
    for (int k = 0; k < 4; ++k) { p.toa_i[k] = op.toa_i[k]; p.toa_s[k] = op.toa_s[k]; }
    for (int k = 0; k < 8; ++k) { p.srf_i[k] = op.srf_i[k]; p.srf_s[k] = op.srf_s[k]; }
    p.gph = op.gph; p.n_seconds = op.n_seconds;
    p.partial = partial; p.sums = sums; p.ratio = ratio; p.n_blocks = n_blocks;
    return p;
  }
  void tracer_op(const PostOp& op, float* y, hipStream_t stream) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(fix_tracer_kernel, dim3(n_blocks), dim3(256), 0, stream, y, h * w, op.n_tr, op.tr_inds, op.tr_lo, op.tr_hi,
                       op.denorm ? mean_out : nullptr, op.denorm ? std_out : nullptr);
    WX_HIP(hipGetLastError());
  }
  // a fixer in two halves: the (local) integrals -> sums[4] ...
  void reduce_op(const PostOp& op, const float* x, float* y, hipStream_t stream) {
    const FixParams p = fix_params(op, x, y);
    if (n_blocks > 0) hipLaunchKernelGGL(fix_reduce_kernel, dim3(n_blocks), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(fix_sum_kernel, dim3(1), dim3(256), 0, stream, p);
    WX_HIP(hipGetLastError());
  }
  // ... and, once sums[4] hold the integrals over the whole globe, the ratio and the correction
  void finish_op(const PostOp& op, const float* x, float* y, hipStream_t stream) {
    const FixParams p = fix_params(op, x, y);
    hipLaunchKernelGGL(fix_ratio_kernel, dim3(1), dim3(1), 0, stream, p);
    if (n_blocks > 0) hipLaunchKernelGGL(fix_apply_kernel, dim3(n_blocks), dim3(256), 0, stream, p);
    WX_HIP(hipGetLastError());
  }
  int n_fixers() const { int n = 0; for (const PostOp& op : ops) n += op.kind != 0; return n; }
  void apply(const float* x, float* y, hipStream_t stream) {
    WX_HIP(hipSetDevice(device));
    if (h != h_full) throw std::runtime_error("wx_post_apply: this block covers a latitude band; it only runs inside a lat-band engine");
    for (const PostOp& op : ops) {
      if (op.kind == 0) { tracer_op(op, y, stream); continue; }
      reduce_op(op, x, y, stream);
      finish_op(op, x, y, stream);
    }
  }
};

}  // namespace wx
