// costmap_warp.hip — rigid global→vehicle-frame costmap warp for gfx950 (MI355X).
//
// Reference: the per-cell loop of LocalCostmap::odomCallback (M/src/local_costmap.cpp:242-264) over the
// grid_map index conventions (G/grid_map_core/src/GridMapMath.cpp:114-156, GridMap.cpp:45-62,160-166):
//   destination cell (i,j) → centre C = (pos + len/2 - res/2) - res·(i,j)
//   → g = Rot(theta)·C + (Vx,Vy) → source index trunc(-((g - len_g/2) - pos_g)/res_g), valid iff inside the source map
//   → dst = src[index];  bbox(cell) > 90 overrides.
// HBM-bound gather/scatter: 4 B read + 4 B write per destination cell (+4 B with the bbox layer).  Layout: float32
// column-major (i fastest).  A workgroup owns a 64(i)×16(j) destination tile: lanes run along i, so destination
// stores (and the bbox loads) are 256-B coalesced per wavefront, and the source footprint of a tile is a small rotated
// rectangle that stays in L1/L2.  Tiles are dealt to XCDs in contiguous runs so neighbouring tiles (which share
// source cache lines) share an L2.
// The index arithmetic is fp64 with contraction OFF and IEEE division so that indices are bit-identical to the
// reference's (an fma here would move cells that sit on a source-cell boundary).
#include <math.h>
#include <stdlib.h>

#include "cilqr_internal.h"

namespace cilqr {

namespace {

constexpr int TILE_I = 64;
constexpr int TILE_J = 16;
constexpr int NTHREADS = 256;  // 4 wavefronts: wave w handles j = w, w+4, w+8, w+12 of the tile

// -(t / res) as far as its truncation to an index is concerned.  t·(1/res) differs from the correctly rounded quotient by
// at most a few ulp, so the two truncate alike unless an integer lies within that distance; only then (cell centres that
// map onto a source-cell boundary to within 1e-13 cells) is the IEEE division the reference performs carried out.
__device__ __forceinline__ double neg_quotient_for_trunc(double t, double res, double rres) {
#pragma clang fp contract(off)
  const double q = -(t * rres);
  const double k = __builtin_rint(q);
  if (__builtin_fabs(q - k) > __builtin_fabs(q) * 0x1p-48 + 0x1p-1000) return q;
  return -(t / res);
}

// One destination cell the reference's way (getIndexFromPosition / checkIfPositionWithinMap on the rotated cell centre): the
// source cell (si, sj), or false when the centre lies outside the source map.  tq receives (t_x, t_y) = the centre relative to the
// source map's far corner, the quantity the neighbouring cells' estimates start from (cell_estimated).
__device__ __forceinline__ bool cell_exact(const cilqr_map_geom& sg, double x_og, double y_og, double off_sx, double off_sy, double rres,
                                           int& si, int& sj, double& t_x, double& t_y) {
#pragma clang fp contract(off)
  t_x = (x_og - off_sx) - sg.pos_x;
  t_y = (y_og - off_sy) - sg.pos_y;
  const double nx = neg_quotient_for_trunc(t_x, sg.res, rres);
  const double ny = neg_quotient_for_trunc(t_y, sg.res, rres);
  const double tx = -1.0 * ((x_og - sg.pos_x) - off_sx);
  const double ty = -1.0 * ((y_og - sg.pos_y) - off_sy);
  const bool fin = (nx > -2e9 && nx < 2e9 && ny > -2e9 && ny < 2e9);
  si = fin ? (int)nx : -1;
  sj = fin ? (int)ny : -1;
  return (tx >= 0.0 && ty >= 0.0 && tx < sg.len_x && ty < sg.len_y) && (si >= 0 && sj >= 0 && si < sg.rows && sj < sg.cols);
}

// A NEIGHBOURING cell from an estimate of its (t_x, t_y): within a lane's run of cells the rotated centre advances by a constant
// vector, so the neighbours' t are the first cell's plus multiples of that step — good to a few ulp of the coordinates, i.e.
// ≈ 2^-50 of `mag` (the sum of every magnitude that enters a centre, in source cells).  The estimate decides a cell only when
// its quotient is further than guard = mag·2^-40 from every integer — then the exact arithmetic above truncates to the same
// index and lands on the same side of the map's edges (0 and rows are integers too); otherwise (a centre within 1e-12 of a
// source-cell boundary: aligned grids, or once in ≈ 1e9 cells) false is returned and the caller evaluates the cell exactly.
// ≈ 20 instructions against ≈ 48.  NaN and infinite poses fail the guard test and take the exact path.
__device__ __forceinline__ bool cell_estimated(const cilqr_map_geom& sg, double t_x, double t_y, double nrres, double guard, bool& ok, int& si, int& sj) {
#pragma clang fp contract(off)
  const double qx = t_x * nrres, qy = t_y * nrres;  // nrres = -1/res
  const double dx = __builtin_fabs(qx - __builtin_rint(qx)), dy = __builtin_fabs(qy - __builtin_rint(qy));
  if (!(dx > guard && dy > guard)) return false;
  ok = qx > 0.0 && qy > 0.0 && qx < (double)sg.rows && qy < (double)sg.cols;
  si = ok ? (int)qx : -1;
  sj = ok ? (int)qy : -1;
  return true;
}
// mag·2^-40 (cell_estimated)
__device__ __forceinline__ double estimate_guard(const cilqr_map_geom& sg, const cilqr_map_geom& dg, double vx, double vy, double rres) {
  const double mag = (__builtin_fabs(sg.pos_x) + __builtin_fabs(sg.pos_y) + sg.len_x + sg.len_y + __builtin_fabs(dg.pos_x) + __builtin_fabs(dg.pos_y) +
                      dg.len_x + dg.len_y + __builtin_fabs(vx) + __builtin_fabs(vy)) * rres;
  return mag * 0x1p-40 + 0x1p-1000;
}

__global__ __launch_bounds__(NTHREADS) void warp_kernel(WarpArgs a, int tiles_i, int n_tiles) {
#pragma clang fp contract(off)
  // XCD-aware remap: workgroup ids are dealt round-robin over the 8 XCDs (speed only, never correctness).
  int bid = blockIdx.x;
  {
    const int per = n_tiles / 8;
    if (per > 0 && bid < per * 8) bid = (bid % 8) * per + bid / 8;
  }
  const int ti = bid % tiles_i, tj = bid / tiles_i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = ti * TILE_I + lane;
  const int drows = a.dg.rows, dcols = a.dg.cols;
  const bool in_i = i < drows;  // lanes past the last row stay in the wavefront for the shuffle reduction below, idle

  const double off_dx = 0.5 * a.dg.len_x - 0.5 * a.dg.res, off_dy = 0.5 * a.dg.len_y - 0.5 * a.dg.res;
  const double Cx = (a.dg.pos_x + off_dx) + a.dg.res * (double)(-i);
  const double off_sx = 0.5 * a.sg.len_x, off_sy = 0.5 * a.sg.len_y;
  const double cxc = Cx * a.cos_t, cxs = Cx * a.sin_t;
  const double rres = 1.0 / a.sg.res, nrres = -rres;
  const double guard = estimate_guard(a.sg, a.dg, a.vx, a.vy, rres);
  // the wavefront's columns are four apart: the centre advances by (+4 res sin, -4 res cos) from one to the next
  const double step_x = 4.0 * a.dg.res * a.sin_t, step_y = -4.0 * a.dg.res * a.cos_t;
  double t0x = 0.0, t0y = 0.0;
  unsigned long long oob = 0;
#pragma unroll
  for (int jj = 0; jj < TILE_J / 4; ++jj) {
    const int j = tj * TILE_J + wave + 4 * jj;
    if (j >= dcols || !in_i) break;
    bool ok;
    int si, sj;
    bool have = false;
    if (jj > 0) have = cell_estimated(a.sg, t0x + (double)jj * step_x, t0y + (double)jj * step_y, nrres, guard, ok, si, sj);
    if (!have) {  // the first column of the run, and any cell whose estimate is too close to a source-cell boundary
      const double Cy = (a.dg.pos_y + off_dy) + a.dg.res * (double)(-j);
      const double x_og = (cxc - Cy * a.sin_t) + a.vx;
      const double y_og = (cxs + Cy * a.cos_t) + a.vy;
      double tx_, ty_;
      ok = cell_exact(a.sg, x_og, y_og, off_sx, off_sy, rres, si, sj, tx_, ty_);
      if (jj == 0) { t0x = tx_; t0y = ty_; }
    }
    const size_t lin = (size_t)j * drows + i;
    float v;
    if (ok) {
      v = a.src[(size_t)sj * a.sg.rows + si];
    } else {
      v = __builtin_nanf("");
      ++oob;
    }
    if (a.bbox) {
      const float bb = a.bbox[lin];
      if (bb > 90.0f) v = bb;
    }
    a.dst[lin] = v;
  }
  if (a.n_oob) {
    // one atomic per wavefront
    for (int o = 32; o > 0; o >>= 1) oob += __shfl_xor(oob, o, 64);
    if (lane == 0 && oob) atomicAdd(a.n_oob, oob);
  }
}

// ---- K frames per launch, four consecutive rows per lane -------------------------------------------------------------------------
// One 1024² frame is 8 MiB of traffic: ≈1 µs at HBM speed against ≈5 µs of launch and ramp — a single frame cannot be
// bandwidth-bound.  A pose stream (the node's pose-noise candidates, a backlog of odometry ticks at 30 Hz) is warped K frames per
// launch instead: blockIdx.y = frame, pose (vx, vy, sin, cos) from a device table, destination frames back to back.  A lane owns
// four consecutive rows i of one column: ONE 16-byte store (and one 16-byte bbox load) per lane and column, a whole KiB per
// wavefront instruction, the four source cells gathered separately (they are neighbours in the source: same or adjacent
// lines).  The per-cell index arithmetic is the single-frame kernel's, bit for bit.
constexpr int VT_J = 8;    // columns per tile: wave w handles j = w, w + 4

// ROWS consecutive rows per lane (4 or 8): ROWS / 4 16-byte stores per lane and column; one cell of a run exact, the others by
// estimate; a cell that lands in the source cell of the one before it (two of four do at 0.1 m against 0.2 m resolution, whatever
// the rotation) takes its value without another gather.
// REUSE: that reuse — worth 6-8 % where frames are many (K ≥ 4); for ONE frame, which is a latency chain and not work, the
// comparison in front of every gather costs 1.5 µs of 6 (rocprofv3: 7.7 against 6.2 µs), so single frames run without it.
template <int ROWS, bool REUSE>
__global__ __launch_bounds__(NTHREADS) void warp_batch_kernel(WarpBatchArgs a, int tiles_i) {
#pragma clang fp contract(off)
  constexpr int VT_I = 64 * ROWS;  // rows per tile
  const int frame = blockIdx.y;
  const double* pose = a.poses ? a.poses + 4 * (size_t)frame : a.pose0;  // (wave-uniform; a single frame carries its pose in the arguments)
  const double vx = pose[0], vy = pose[1], sin_t = pose[2], cos_t = pose[3];
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = ti * VT_I + ROWS * lane;
  const int drows = a.dg.rows, dcols = a.dg.cols;
  const size_t frame_cells = (size_t)drows * dcols;
  float* dst = a.dst + frame * frame_cells;
  const double off_dx = 0.5 * a.dg.len_x - 0.5 * a.dg.res, off_dy = 0.5 * a.dg.len_y - 0.5 * a.dg.res;
  const double off_sx = 0.5 * a.sg.len_x, off_sy = 0.5 * a.sg.len_y;
  const double rres = 1.0 / a.sg.res, nrres = -rres;
  const double guard = estimate_guard(a.sg, a.dg, vx, vy, rres);
  // a lane's rows: the centre advances by (-res cos, -res sin) from one row to the next
  const double step_x = -a.dg.res * cos_t, step_y = -a.dg.res * sin_t;
  unsigned long long oob = 0;
  const bool in_i = i0 < drows;  // drows is a multiple of ROWS on this path: a lane's rows are all inside or all outside
#pragma unroll
  for (int jj = 0; jj < VT_J / 4; ++jj) {
    const int j = tj * VT_J + wave + 4 * jj;
    if (j >= dcols || !in_i) break;
    const double Cy = (a.dg.pos_y + off_dy) + a.dg.res * (double)(-j);
    const double cys = Cy * sin_t, cyc = Cy * cos_t;
    float v[ROWS];
    double t0x = 0.0, t0y = 0.0;
    int psi = -2, psj = -2;  // the source cell of the cell before (-2: none)
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
      bool ok;
      int si, sj;
      bool have = false;
      if (k > 0) have = cell_estimated(a.sg, t0x + (double)k * step_x, t0y + (double)k * step_y, nrres, guard, ok, si, sj);
      if (!have) {  // the first row of the run, and any cell whose estimate is too close to a source-cell boundary
        const double Cx = (a.dg.pos_x + off_dx) + a.dg.res * (double)(-(i0 + k));
        const double x_og = (Cx * cos_t - cys) + vx;
        const double y_og = (Cx * sin_t + cyc) + vy;
        double tx_, ty_;
        ok = cell_exact(a.sg, x_og, y_og, off_sx, off_sy, rres, si, sj, tx_, ty_);
        if (k == 0) { t0x = tx_; t0y = ty_; }
      }
      if (!ok) {
        v[k] = __builtin_nanf("");
        ++oob;
        psi = -2;
      } else if (REUSE && k > 0 && si == psi && sj == psj) {
        v[k] = v[k - 1];
      } else {
        v[k] = a.src[(size_t)sj * a.sg.rows + si];
        psi = si; psj = sj;
      }
    }
    const size_t lin = (size_t)j * drows + i0;
#pragma unroll
    for (int q = 0; q < ROWS / 4; ++q) {
      if (a.bbox) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bbox + lin + 4 * q);
        if (bb.x > 90.0f) v[4 * q] = bb.x;
        if (bb.y > 90.0f) v[4 * q + 1] = bb.y;
        if (bb.z > 90.0f) v[4 * q + 2] = bb.z;
        if (bb.w > 90.0f) v[4 * q + 3] = bb.w;
      }
      *reinterpret_cast<float4*>(dst + lin + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
  }
  if (a.n_oob) {
    for (int o = 32; o > 0; o >>= 1) oob += __shfl_xor(oob, o, 64);
    if (lane == 0 && oob) atomicAdd(a.n_oob + frame, oob);
  }
}

// ---- the same frames through LDS: a coalesced 2-D tiled kernel (experiment, see launch_warp_batch) ---------------------------------
// The gathers of warp_batch_kernel go through the texture path one cache line per distinct source cell and lane.  Here a workgroup owns a compact destination tile — 64 rows × 32 columns, 6.4 m ×
// 3.2 m at the node's 0.1 m — whose source footprint is a small rotated rectangle: its bounding box in source cells (from the
// tile's four corner centres, widened by two cells) is loaded ONCE into LDS with coalesced row segments, and every destination
// cell then reads its source cell from LDS.  Index arithmetic and results are those of the other kernels, bit for bit; a source
// cell outside the staged box (cannot happen by construction; guarded all the same) is read from global memory.
constexpr int LT_I = 64, LT_J = 32;  // destination tile: rows × columns; 256 threads = 16 row groups of 4 rows × 16 columns, two columns each

__global__ __launch_bounds__(NTHREADS) void warp_tile_kernel(WarpBatchArgs a, int tiles_i, int lds_cells) {
#pragma clang fp contract(off)
  extern __shared__ float tile_src[];
  const int frame = blockIdx.y;
  const double* pose = a.poses ? a.poses + 4 * (size_t)frame : a.pose0;
  const double vx = pose[0], vy = pose[1], sin_t = pose[2], cos_t = pose[3];
  const int ti = blockIdx.x % tiles_i, tj = blockIdx.x / tiles_i;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int drows = a.dg.rows, dcols = a.dg.cols;
  const size_t frame_cells = (size_t)drows * dcols;
  float* dst = a.dst + frame * frame_cells;
  const double off_dx = 0.5 * a.dg.len_x - 0.5 * a.dg.res, off_dy = 0.5 * a.dg.len_y - 0.5 * a.dg.res;
  const double off_sx = 0.5 * a.sg.len_x, off_sy = 0.5 * a.sg.len_y;
  const double rres = 1.0 / a.sg.res, nrres = -rres;
  const double guard = estimate_guard(a.sg, a.dg, vx, vy, rres);
  const double step_x = -a.dg.res * cos_t, step_y = -a.dg.res * sin_t;

  // bounding box of the tile's footprint in source cells: the map is affine, so the extremes lie at the corner centres
  int si_lo, si_hi, sj_lo, sj_hi;
  {
    const int ia = ti * LT_I, ib = min(ia + LT_I, drows) - 1, ja = tj * LT_J, jb = min(ja + LT_J, dcols) - 1;
    double qx_lo = 1e300, qx_hi = -1e300, qy_lo = 1e300, qy_hi = -1e300;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double Cx = (a.dg.pos_x + off_dx) + a.dg.res * (double)(-((c & 1) ? ib : ia));
      const double Cy = (a.dg.pos_y + off_dy) + a.dg.res * (double)(-((c & 2) ? jb : ja));
      const double qx = (((Cx * cos_t - Cy * sin_t) + vx - off_sx) - a.sg.pos_x) * nrres;
      const double qy = (((Cx * sin_t + Cy * cos_t) + vy - off_sy) - a.sg.pos_y) * nrres;
      qx_lo = __builtin_fmin(qx_lo, qx); qx_hi = __builtin_fmax(qx_hi, qx);
      qy_lo = __builtin_fmin(qy_lo, qy); qy_hi = __builtin_fmax(qy_hi, qy);
    }
    // (NaN / infinite poses: the comparisons below fail and the box is empty — every cell then takes the global path or is out of range)
    const bool fin = qx_lo > -2e9 && qx_hi < 2e9 && qy_lo > -2e9 && qy_hi < 2e9;
    si_lo = fin ? max((int)__builtin_floor(qx_lo) - 2, 0) : 0;
    si_hi = fin ? min((int)__builtin_floor(qx_hi) + 2, a.sg.rows - 1) : -1;
    sj_lo = fin ? max((int)__builtin_floor(qy_lo) - 2, 0) : 0;
    sj_hi = fin ? min((int)__builtin_floor(qy_hi) + 2, a.sg.cols - 1) : -1;
  }
  int W = si_hi - si_lo + 1, H = sj_hi - sj_lo + 1;
  if (W <= 0 || H <= 0 || (long)W * H > lds_cells) { W = 0; H = 0; }  // nothing staged: the guard below sends every read to global memory
  for (int h = wave; h < H; h += NTHREADS / 64) {
    const float* row = a.src + (size_t)(sj_lo + h) * a.sg.rows + si_lo;
    for (int w = lane; w < W; w += 64) tile_src[h * W + w] = row[w];
  }
  __syncthreads();

  const int rg = tid & 15, cj = tid >> 4;
  const int i0 = ti * LT_I + 4 * rg;
  unsigned long long oob = 0;
  if (i0 < drows) {  // (drows is a multiple of 4 on this path)
#pragma unroll
    for (int jj = 0; jj < LT_J / 16; ++jj) {
      const int j = tj * LT_J + cj + 16 * jj;
      if (j >= dcols) break;
      const double Cy = (a.dg.pos_y + off_dy) + a.dg.res * (double)(-j);
      const double cys = Cy * sin_t, cyc = Cy * cos_t;
      float v[4];
      double t0x = 0.0, t0y = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool ok;
        int si, sj;
        bool have = false;
        if (k > 0) have = cell_estimated(a.sg, t0x + (double)k * step_x, t0y + (double)k * step_y, nrres, guard, ok, si, sj);
        if (!have) {
          const double Cx = (a.dg.pos_x + off_dx) + a.dg.res * (double)(-(i0 + k));
          const double x_og = (Cx * cos_t - cys) + vx;
          const double y_og = (Cx * sin_t + cyc) + vy;
          double tx_, ty_;
          ok = cell_exact(a.sg, x_og, y_og, off_sx, off_sy, rres, si, sj, tx_, ty_);
          if (k == 0) { t0x = tx_; t0y = ty_; }
        }
        if (!ok) {
          v[k] = __builtin_nanf("");
          ++oob;
        } else {
          const int u = si - si_lo, w = sj - sj_lo;
          v[k] = ((unsigned)u < (unsigned)W && (unsigned)w < (unsigned)H) ? tile_src[w * W + u] : a.src[(size_t)sj * a.sg.rows + si];
        }
      }
      const size_t lin = (size_t)j * drows + i0;
      if (a.bbox) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bbox + lin);
        if (bb.x > 90.0f) v[0] = bb.x;
        if (bb.y > 90.0f) v[1] = bb.y;
        if (bb.z > 90.0f) v[2] = bb.z;
        if (bb.w > 90.0f) v[3] = bb.w;
      }
      *reinterpret_cast<float4*>(dst + lin) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (a.n_oob) {
    for (int o = 32; o > 0; o >>= 1) oob += __shfl_xor(oob, o, 64);
    if (lane == 0 && oob) atomicAdd(a.n_oob + frame, oob);
  }
}

}  // namespace

hipError_t launch_warp_batch(const WarpBatchArgs& a, int K, hipStream_t stream) {
  if (K <= 0) return hipSuccess;
  if (a.dg.rows % 4 != 0) return hipErrorInvalidValue;  // the caller falls back to per-frame launches
  // The LDS-tiled kernel is an experiment, OFF unless CILQR_WARP_LDS=1: bit-exact, but slower than the gather kernel below at every
  // K (profiles/r03_warp.txt: 3.4 against 2.4 µs per frame at K = 16, 2.7 against 1.9 at K = 64) — the texture path's gathers from an
  // L2-resident patch were not what bounds the kernel, and the staging (box, loads, barrier, four workgroups' worth of LDS) is not free.
  {
    const double ratio = a.dg.res / a.sg.res;
    const double side = ceil(sqrt((double)(LT_I * LT_I + LT_J * LT_J)) * ratio) + 6.0;
    const char* sw = getenv("CILQR_WARP_LDS");
    if (sw && atoi(sw) != 0 && side * side * sizeof(float) <= 48.0 * 1024.0) {
      const int cells = (int)(side * side);
      const int ti = (a.dg.rows + LT_I - 1) / LT_I, tj = (a.dg.cols + LT_J - 1) / LT_J;
      hipLaunchKernelGGL(warp_tile_kernel, dim3(ti * tj, K), dim3(NTHREADS), (size_t)cells * sizeof(float), stream, a, ti, cells);
      return hipGetLastError();
    }
  }
  const int tiles_j = (a.dg.cols + VT_J - 1) / VT_J;
  // Four rows per lane.  Eight (CILQR_WARP_ROWS=8: one exact cell in eight) measured slower at every K — 2.83 against 2.41 µs per
  // frame at K = 16, 10.6 against 8.4 µs for one frame (profiles/r03_warp.txt): half the workgroups, and the arithmetic was not
  // what bounds the kernel.
  const int rows8 = getenv("CILQR_WARP_ROWS") ? atoi(getenv("CILQR_WARP_ROWS")) : 4;
  if (a.dg.rows % 8 == 0 && rows8 == 8) {
    const int tiles_i = (a.dg.rows + 511) / 512;
    hipLaunchKernelGGL((warp_batch_kernel<8, true>), dim3(tiles_i * tiles_j, K), dim3(NTHREADS), 0, stream, a, tiles_i);
  } else {
    const int tiles_i = (a.dg.rows + 255) / 256;
    if (K >= 4) hipLaunchKernelGGL((warp_batch_kernel<4, true>), dim3(tiles_i * tiles_j, K), dim3(NTHREADS), 0, stream, a, tiles_i);
    else hipLaunchKernelGGL((warp_batch_kernel<4, false>), dim3(tiles_i * tiles_j, K), dim3(NTHREADS), 0, stream, a, tiles_i);
  }
  return hipGetLastError();
}

hipError_t launch_warp(const WarpArgs& a, hipStream_t stream) {
  if (a.dg.rows % 4 == 0) {  // four rows per lane, 16-byte stores: a quarter of the workgroups to dispatch, three of four cells by estimate
    WarpBatchArgs b;
    b.src = a.src; b.dst = a.dst; b.bbox = a.bbox; b.n_oob = a.n_oob; b.poses = nullptr;
    b.pose0[0] = a.vx; b.pose0[1] = a.vy; b.pose0[2] = a.sin_t; b.pose0[3] = a.cos_t;
    b.sg = a.sg; b.dg = a.dg;
    return launch_warp_batch(b, 1, stream);
  }
  const int tiles_i = (a.dg.rows + TILE_I - 1) / TILE_I, tiles_j = (a.dg.cols + TILE_J - 1) / TILE_J;
  const int n_tiles = tiles_i * tiles_j;
  if (n_tiles <= 0) return hipSuccess;
  hipLaunchKernelGGL(warp_kernel, dim3(n_tiles), dim3(NTHREADS), 0, stream, a, tiles_i, n_tiles);
  return hipGetLastError();
}

}  // namespace cilqr
