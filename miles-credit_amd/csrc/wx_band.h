// Lat-band sharding of ONE forecast over n ranks (SURVEY.md §8(e) mode 2; reference protocol:
// credit/parallel/domain.py:25-64 shard_spatial / gather_spatial, credit/domain_parallel/layers.py:72-110 conv halos,
// :507-522 GroupNorm all-reduce, halo_exchange.py:45-79 neighbour rows, zero fill at the poles).
//
// This header is the HOST-ONLY planning half: which rows of which map every rank owns, and, for every exchange step
// of a forecast step, which row runs travel between which ranks.  It is plain C++ (no HIP) so that the CPU tests can
// exercise it through the C ABI (wx_band_plan_*).  The engine (wx_engine.hip) executes the plan; the transport
// (RCCL / gloo / in-process copies) only ever sees "send these bytes of my staging buffer to rank p".
//
// Differences from the reference, on purpose:
//  * the reference splits the padded grid into n EQUAL bands and runs every window attention inside the band, which
//    changes the long (dilated) attention and fails when a band is not a multiple of the window.  Here the split is
//    window-aligned (ragged where the window count does not divide by n, possibly empty at the deepest stage), and the
//    long attention is exact: rows are redistributed so that every rank holds whole dilated windows
//    (phase i of G = H / wsz owns rows {i + G k}), attended, and sent back -- one all-to-all each way.
//  * one halo exchange per convolution input (the reference issues one per CrossEmbed branch: 4 at stage 0).
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace wx {

struct BandSeg {      // `nrows` rows: local row `src_row` of rank `peer`  ->  local row `dst_row` of the receiving rank
  int peer;           // source rank; -1 = rows outside the global map: zero fill
  int src_row, dst_row, nrows;
};

struct BandExchange {
  std::string name;
  int src_buf = -1, dst_buf = -1;            // engine buffer ids (wx_engine.hip: enum BandBuf)
  int stage = 0;                             // which geometry the buffers have
  int64_t row_bytes = 0;                     // packed bytes of one row
  std::vector<std::vector<BandSeg>> recv;    // [rank] -> what that rank receives, in destination-row order
};

struct BandMsg { int peer; int64_t offset, bytes; };   // one contiguous block of a staging buffer

// balanced contiguous split of `units` items over n ranks: start of rank r
inline int band_split(int units, int n, int r) { return (int)((int64_t)units * r / n); }

struct BandGeom {
  int n = 1;
  // per stage: short layout (window-aligned rows) and long layout (phases of the dilated windows)
  int sh[4] = {0, 0, 0, 0}, wl[4] = {1, 1, 1, 1}, wg[4] = {1, 1, 1, 1};
  std::vector<int> ps[4];   // [n+1] first owned ROW of each rank in the short layout
  std::vector<int> pl[4];   // [n+1] first owned PHASE of each rank in the long layout (rows = phases * wg)
  std::vector<int> po;      // [n+1] first owned row of the input / output grid
  int out_h = 0;

  void init(int n_, const int sh_[4], const int wl_[4], const int wg_[4]) {
    n = n_;
    for (int s = 0; s < 4; ++s) {
      sh[s] = sh_[s]; wl[s] = wl_[s]; wg[s] = wg_[s];
      if (sh[s] % wl[s] || sh[s] % wg[s]) throw std::runtime_error("band: stage map not divisible by the window");
      ps[s].resize(n + 1);
      pl[s].resize(n + 1);
      for (int r = 0; r <= n; ++r) {
        ps[s][r] = band_split(sh[s] / wl[s], n, r) * wl[s];
        pl[s][r] = band_split(sh[s] / wg[s], n, r);
      }
    }
  }
  int rows_short(int s, int r) const { return ps[s][r + 1] - ps[s][r]; }
  int rows_long(int s, int r) const { return (pl[s][r + 1] - pl[s][r]) * wg[s]; }
  int max_rows(int s, int r) const { return std::max(rows_short(s, r), rows_long(s, r)); }
};

using BandOwner = std::function<void(int g, int* rank, int* local)>;  // global row -> (owner, its local row)

// Row runs that bring `need[k]` (a global row, or -1 / out of range for zero) to destination local row dst0 + k.
inline std::vector<BandSeg> band_gather(const std::vector<int>& need, int dst0, int global_rows, const BandOwner& owner) {
  std::vector<BandSeg> out;
  for (size_t k = 0; k < need.size(); ++k) {
    int rank = -1, local = 0;
    if (need[k] >= 0 && need[k] < global_rows) owner(need[k], &rank, &local);
    if (!out.empty()) {
      BandSeg& b = out.back();
      if (b.peer == rank && b.dst_row + b.nrows == dst0 + (int)k && (rank < 0 || b.src_row + b.nrows == local)) {
        ++b.nrows;
        continue;
      }
    }
    out.push_back(BandSeg{rank, local, dst0 + (int)k, 1});
  }
  return out;
}

inline BandOwner band_owner_rows(const std::vector<int>& starts, int local0 = 0) {  // contiguous row partition
  return [starts, local0](int g, int* rank, int* local) {
    const int r = (int)(std::upper_bound(starts.begin(), starts.end(), g) - starts.begin()) - 1;
    *rank = r;
    *local = g - starts[r] + local0;
  };
}
// long layout: global row g = phase i + G * k lives at local row (i - first_phase) * wg + k of the rank owning phase i
inline BandOwner band_owner_long(const std::vector<int>& phase_starts, int G, int wg) {
  return [phase_starts, G, wg](int g, int* rank, int* local) {
    const int i = g % G, k = g / G;
    const int r = (int)(std::upper_bound(phase_starts.begin(), phase_starts.end(), i) - phase_starts.begin()) - 1;
    *rank = r;
    *local = (i - phase_starts[r]) * wg + k;
  };
}

// The messages of rank `me` for one exchange: both sides walk the same global table in the same order
// (destination rank ascending, then its segment order), so offsets agree without any negotiation.
inline void band_messages(const BandExchange& x, int me, std::vector<BandMsg>* sends, std::vector<BandMsg>* recvs) {
  sends->clear();
  recvs->clear();
  const int n = (int)x.recv.size();
  int64_t off = 0;
  for (int p = 0; p < n; ++p) {  // what I send: segments of every OTHER rank p whose source is me
    if (p == me) continue;
    int64_t bytes = 0;
    for (const BandSeg& s : x.recv[p])
      if (s.peer == me) bytes += (int64_t)s.nrows * x.row_bytes;
    if (bytes) sends->push_back(BandMsg{p, off, bytes});
    off += bytes;
  }
  off = 0;
  for (int r = 0; r < n; ++r) {  // what I receive, grouped by source rank
    if (r == me) continue;
    int64_t bytes = 0;
    for (const BandSeg& s : x.recv[me])
      if (s.peer == r) bytes += (int64_t)s.nrows * x.row_bytes;
    if (bytes) recvs->push_back(BandMsg{r, off, bytes});
    off += bytes;
  }
}
inline int64_t band_send_bytes(const BandExchange& x, int me) {
  int64_t b = 0;
  for (size_t p = 0; p < x.recv.size(); ++p)
    if ((int)p != me)
      for (const BandSeg& s : x.recv[p])
        if (s.peer == me) b += (int64_t)s.nrows * x.row_bytes;
  return b;
}
inline int64_t band_recv_bytes(const BandExchange& x, int me) {
  int64_t b = 0;
  for (const BandSeg& s : x.recv[me])
    if (s.peer >= 0 && s.peer != me) b += (int64_t)s.nrows * x.row_bytes;
  return b;
}

// ------------------------------------------------------------------------------------------------------------------
// The exchange schedule of one forecast step (legacy CrossFormer architecture).  Buffer ids are resolved by the engine.
enum BandBuf {
  BB_X_OWN = 0,   // caller's input band   [C_in][own rows][W] fp32
  BB_X_NEED,      // engine: input rows the band's padded patch needs [C_in][need rows][W] fp32
  BB_STREAM_S,    // stage stream, short layout (window-aligned rows, 1 halo row above / below for stages 0-2)
  BB_STREAM_L,    // stage stream, long layout (whole dilated windows)
  BB_EMB_IN,      // CrossEmbed input rows of the previous stage (with the conv's halo)
  BB_DEC_SRC,     // decoder level input at its own stage partition: x3 (level 0) or the concat buffer of stage si
  BB_DEC_IN,      // ... gathered to the rows the output partition needs
  BB_SCUT,        // decoder shortcut (ConvTranspose output), 1 halo row
  BB_TB,          // decoder intermediate, 1 halo row
  BB_CAT0,        // full concat buffer of stage 0 (input of the k4s2 ConvTranspose), 1 halo row
  BB_DEC,         // decoder output map, 1 halo row
  BB_GN_ACC,      // GroupNorm (sum, sum sq) of this rank: one "row"
  BB_GN_ALL,      // ... of every rank: n rows
  BB_PS4,         // wxformer head: pixel-shuffled map before the final 3x3 conv, 1 halo row
  BB_FIX_ACC,     // post-block fixer: this rank's 4 global-integral partial sums (one row of 32 bytes)
  BB_FIX_ALL,     // ... of every rank
};

struct BandModel {
  int n = 1;
  int C_in = 0, H = 0, W = 0;            // input grid
  int p0 = 0, p1 = 0;                    // pad_lat (0 when padding is off)
  int Hp = 0, halo = 0;                  // padded rows, stage-0 patch halo (rows of the padded grid)
  int stride[4] = {2, 2, 2, 2};
  int emb_lo[4] = {0, 0, 0, 0}, emb_hi[4] = {0, 0, 0, 0};  // rows a CrossEmbed output row reaches above / below stride*j .. stride*j+stride-1
  int sh[4], sw[4], wl[4], wg[4], depth[4], dim[4];
  int elem = 2;                          // sizeof(T)
  int up_cout[3] = {0, 0, 0};            // decoder level output channels
  int Hd = 0, Wd = 0, Hu = 0, Ho = 0, off_y = 0, interp = 0, ld_dec = 0;
  int wxformer = 0, cpad4 = 0;           // PixelShuffle decoder (wxformer/crossformer.py:137-162, :817-830)
  int n_fix = 0;                         // conservation fixers of the attached post block (one sum exchange each)
};

struct BandPlan {
  BandModel m;
  BandGeom g;
  std::vector<BandExchange> xs;
  std::vector<int> x_need_lo, x_need_hi;   // [rank] input rows [lo, hi) the rank's padded patch reads
  std::vector<int> pad_lo, pad_hi;         // [rank] padded rows [lo, hi) it packs

  // vertical source rows of output row oy (tail_kernel: ATen bilinear, align_corners = False)
  void out_src(int oy, int* r0, int* r1) const {
    if (m.interp && m.Hu != m.Ho) {
      const float sc = (float)m.Hu / (float)m.Ho;
      float src = sc * ((float)oy + 0.5f) - 0.5f;
      src = src < 0.f ? 0.f : src;
      int a = (int)src;
      if (a > m.Hu - 1) a = m.Hu - 1;
      *r0 = a;
      *r1 = a + 1 > m.Hu - 1 ? m.Hu - 1 : a + 1;
    } else {
      *r0 = *r1 = oy;
    }
  }
  // wxformer decoder level with output rows [a, b) of a map of m.sh[so] rows: input rows [j0, j1) whose pixel-shuffled
  // rows cover [max(a-1, 0), min(b+1, sh)) -- what the "sharp" 3x3 conv of the owned rows reads
  void dec_ps_rows(int so, int a, int b, int* j0, int* j1) const {
    const int pa = std::max(a - 1, 0), pb = std::min(b + 1, m.sh[so]);
    *j0 = pa / 2;
    *j1 = (pb + 1) / 2;
  }
  int src_row_of_padded(int gp) const {  // boundary_padding.py:50-72: the pole pads mirror rows 0..p-1 / H-p..H-1
    if (gp < m.p0) return m.p0 - 1 - gp;
    if (gp >= m.p0 + m.H) return m.H - 1 - (gp - m.p0 - m.H);
    return gp - m.p0;
  }

  void add_gather(const std::string& name, int src, int dst, int stage, int64_t row_bytes, int global_rows,
                  const std::function<std::vector<int>(int)>& need_of, const std::function<int(int)>& dst0_of,
                  const BandOwner& owner) {
    BandExchange x;
    x.name = name; x.src_buf = src; x.dst_buf = dst; x.stage = stage; x.row_bytes = row_bytes;
    x.recv.resize(m.n);
    for (int r = 0; r < m.n; ++r) x.recv[r] = band_gather(need_of(r), dst0_of(r), global_rows, owner);
    xs.push_back(std::move(x));
  }
  // 1 halo row above and below the owned rows [starts[r], starts[r+1]) of a haloed buffer (owned row 0 = local row 1)
  void add_halo(const std::string& name, int buf, int stage, int64_t row_bytes, int global_rows, const std::vector<int>& starts) {
    BandExchange x;
    x.name = name; x.src_buf = buf; x.dst_buf = buf; x.stage = stage; x.row_bytes = row_bytes;
    x.recv.resize(m.n);
    const BandOwner owner = band_owner_rows(starts, 1);
    for (int r = 0; r < m.n; ++r) {
      const int a = starts[r], b = starts[r + 1];
      if (b == a) continue;  // owns nothing at this stage
      auto top = band_gather({a - 1}, 0, global_rows, owner);
      auto bot = band_gather({b}, b - a + 1, global_rows, owner);
      x.recv[r] = top;
      x.recv[r].insert(x.recv[r].end(), bot.begin(), bot.end());
    }
    xs.push_back(std::move(x));
  }

  void build(const BandModel& model) {
    m = model;
    g.init(m.n, m.sh, m.wl, m.wg);
    const int n = m.n;
    for (int r = 0; r < n; ++r)
      if (g.rows_short(0, r) <= 0) throw std::runtime_error("band: more ranks than window rows at stage 0");
    // ---- output / input row ownership follows the decoder rows of stage 0: out row oy belongs to whoever owns dec row r0
    g.out_h = m.Ho;
    g.po.assign(n + 1, 0);
    {
      int r = 0;
      for (int oy = 0; oy < m.Ho; ++oy) {
        int r0, r1;
        out_src(oy, &r0, &r1);
        const int drow = r0 + m.off_y;                      // decoder row (2 per stage-0 row)
        while (r < n - 1 && drow >= 2 * g.ps[0][r + 1]) { ++r; g.po[r] = oy; }
      }
      for (++r; r <= n; ++r) g.po[r] = m.Ho;
      g.po[n] = m.Ho;
    }
    if (m.Ho != m.H) throw std::runtime_error("band: output rows must equal input rows (rollout feeds y back into x)");
    // ---- padded rows each rank packs, and the input rows those come from
    x_need_lo.assign(n, 0); x_need_hi.assign(n, 0); pad_lo.assign(n, 0); pad_hi.assign(n, 0);
    for (int r = 0; r < n; ++r) {
      const int a = g.ps[0][r], b = g.ps[0][r + 1];
      if (a == b) continue;
      const int lo = std::max(0, m.stride[0] * a - m.halo), hi = std::min(m.Hp, m.stride[0] * b + m.halo);
      pad_lo[r] = lo; pad_hi[r] = hi;
      int xl = m.H, xh = 0;
      for (int gp = lo; gp < hi; ++gp) {
        const int sr = src_row_of_padded(gp);
        xl = std::min(xl, sr); xh = std::max(xh, sr + 1);
      }
      x_need_lo[r] = xl; x_need_hi[r] = xh;
    }
    add_gather("x_rows", BB_X_OWN, BB_X_NEED, 0, (int64_t)m.C_in * m.W * 4, m.H,
               [&](int r) { std::vector<int> v; for (int i = x_need_lo[r]; i < x_need_hi[r]; ++i) v.push_back(i); return v; },
               [](int) { return 0; }, band_owner_rows(g.po));
    // ---- encoder
    for (int s = 0; s < 4; ++s) {
      const int64_t rb = (int64_t)m.sw[s] * m.dim[s] * m.elem;
      if (s > 0) {
        const int st = m.stride[s], lo = m.emb_lo[s], hi = m.emb_hi[s];
        add_gather("embed_in.s" + std::to_string(s), BB_STREAM_S, BB_EMB_IN, s - 1, (int64_t)m.sw[s - 1] * m.dim[s - 1] * m.elem, m.sh[s - 1],
                   [&, s, st, lo, hi](int r) {
                     std::vector<int> v;
                     const int a = g.ps[s][r], b = g.ps[s][r + 1];
                     if (a < b) for (int i = st * a - lo; i < st * b + hi; ++i) v.push_back(i);
                     return v;
                   },
                   [](int) { return 0; }, band_owner_rows(g.ps[s - 1], s - 1 < 3 ? 1 : 0));
      }
      if (m.wg[s] > 1) {
        const int G = m.sh[s] / m.wg[s], wg = m.wg[s];
        for (int d = 0; d < m.depth[s]; ++d) {
          add_gather("to_long.s" + std::to_string(s) + "." + std::to_string(d), BB_STREAM_S, BB_STREAM_L, s, rb, m.sh[s],
                     [&, s, G, wg](int r) {
                       std::vector<int> v;
                       for (int i = g.pl[s][r]; i < g.pl[s][r + 1]; ++i)
                         for (int k = 0; k < wg; ++k) v.push_back(i + G * k);
                       return v;
                     },
                     [](int) { return 0; }, band_owner_rows(g.ps[s], s < 3 ? 1 : 0));
          add_gather("to_short.s" + std::to_string(s) + "." + std::to_string(d), BB_STREAM_L, BB_STREAM_S, s, rb, m.sh[s],
                     [&, s](int r) { std::vector<int> v; for (int i = g.ps[s][r]; i < g.ps[s][r + 1]; ++i) v.push_back(i); return v; },
                     [s](int) { return s < 3 ? 1 : 0; }, band_owner_long(g.pl[s], G, wg));
        }
      }
    }
    // ---- decoder
    for (int i = 0; i < 3; ++i) {
      const int si = 3 - i, so = 2 - i;
      const int cin = i == 0 ? m.dim[3] : 2 * m.dim[si];
      add_gather("dec_in.l" + std::to_string(i), BB_DEC_SRC, BB_DEC_IN, si, (int64_t)m.sw[si] * cin * m.elem, m.sh[si],
                 [&, so](int r) {
                   std::vector<int> v;
                   const int a = g.ps[so][r], b = g.ps[so][r + 1];
                   if (a >= b) return v;
                   if (!m.wxformer) {   // ConvTranspose k2 s2: output rows 2j, 2j+1 from input row j
                     for (int j = a / 2; j < (b + 1) / 2; ++j) v.push_back(j);
                   } else {             // 3x3 conv -> PixelShuffle -> 3x3 "sharp" conv: the shuffled rows a-1 .. b (inside
                     int j0, j1;        // the map) are recomputed locally, each from input rows j-1 .. j+1
                     dec_ps_rows(so, a, b, &j0, &j1);
                     for (int j = j0 - 1; j < j1 + 1; ++j) v.push_back(j);
                   }
                   return v;
                 },
                 [](int) { return 0; }, band_owner_rows(g.ps[si], si < 3 ? 1 : 0));
      const int64_t rb = (int64_t)m.sw[so] * m.up_cout[i] * m.elem;
      for (int j = 0; j < 2; ++j) {
        add_halo((j == 0 ? "halo_scut.l" : "halo_tb.l") + std::to_string(i), j == 0 ? BB_SCUT : BB_TB, so, rb, m.sh[so], g.ps[so]);
        BandExchange x;   // GroupNorm: every rank gets every rank's (sum, sum sq); summed in rank order (deterministic)
        x.name = "gn.l" + std::to_string(i) + "." + std::to_string(j);
        x.src_buf = BB_GN_ACC; x.dst_buf = BB_GN_ALL; x.stage = so; x.row_bytes = (int64_t)2 * m.up_cout[i] * 8;
        x.recv.resize(n);
        for (int r = 0; r < n; ++r)
          for (int q = 0; q < n; ++q) x.recv[r].push_back(BandSeg{q, 0, q, 1});
        xs.push_back(std::move(x));
      }
    }
    add_halo("halo_cat0", BB_CAT0, 0, (int64_t)m.sw[0] * 2 * m.dim[0] * m.elem, m.sh[0], g.ps[0]);
    {
      std::vector<int> dstarts(n + 1);
      for (int r = 0; r <= n; ++r) dstarts[r] = 2 * g.ps[0][r];
      if (m.wxformer) add_halo("halo_ps4", BB_PS4, 0, (int64_t)m.Wd * m.cpad4 * m.elem, m.Hd, dstarts);
      add_halo("halo_dec", BB_DEC, 0, (int64_t)m.Wd * m.ld_dec * m.elem, m.Hd, dstarts);
    }
    for (int k = 0; k < m.n_fix; ++k) {   // global mass / water / energy integrals (gen1.py:280-1030): every rank's 4 sums to everyone
      BandExchange x;
      x.name = "fix." + std::to_string(k);
      x.src_buf = BB_FIX_ACC; x.dst_buf = BB_FIX_ALL; x.stage = 0; x.row_bytes = 32;
      x.recv.resize(n);
      for (int r = 0; r < n; ++r)
        for (int q = 0; q < n; ++q) x.recv[r].push_back(BandSeg{q, 0, q, 1});
      xs.push_back(std::move(x));
    }
  }
};

}  // namespace wx
