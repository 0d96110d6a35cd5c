// Device-side tile helpers shared by the map kernels (K3 sense+update, K4/K5 fusion): lane geometry of a row segment,
// 16-byte cell groups, packed observation bits, and the incremental 11x11 area sums that feed K6.
#pragma once
#include "ippm_internal.h"

#ifdef __HIPCC__

// ---- row/lane geometry -----------------------------------------------------------------------------------------
// A lane owns VEC grid-aligned cells of one row (VEC = 4: one 16-byte access).  Segments narrower than 64 lanes pack
// 64/lpr rows into one wavefront, lpr = next power of two of the group count.
struct RowGeom {
  int y0;      // grid-aligned first column
  int groups;  // VEC-wide groups per row
  int lpr;     // lanes per row (power of two <= 64)
  int rpw;     // rows per wavefront
  int shift;   // log2(lpr)
};
template <int VEC>
__device__ __forceinline__ RowGeom make_geom(int ya, int yb) {
  RowGeom g;
  g.y0 = ya & ~(VEC - 1);
  g.groups = (yb - g.y0 + VEC - 1) / VEC;
  const int gm1 = max(g.groups - 1, 0);
  g.shift = gm1 == 0 ? 0 : min(32 - __clz(gm1), 6);
  g.lpr = 1 << g.shift;
  g.rpw = 64 >> g.shift;
  return g;
}
// Same, but the lanes per row are chosen (8, 16, 32 or 64) to minimise the wavefront-row instructions needed for a segment
// of `rows` rows handled by `row_slots_per_lane_row` = 64/lpr sub-rows: a 90-cell footprint is 23-24 groups, which fills
// 72-75 % of a 32-lane row but 96-100 % of three passes with 8 lanes (8 rows per instruction, 128-byte row segments).
template <int VEC>
__device__ __forceinline__ RowGeom fit_geom(int ya, int yb, int rows, int waves) {
  RowGeom g;
  g.y0 = ya & ~(VEC - 1);
  g.groups = (yb - g.y0 + VEC - 1) / VEC;
  int best_shift = 6, best_cost = 1 << 30;
#pragma unroll
  for (int sh = 6; sh >= 3; --sh) {  // ties go to the wider row (longer contiguous segments)
    const int lpr = 1 << sh, slots = (64 >> sh) * waves;
    const int cost = ((g.groups + lpr - 1) >> sh) * ((rows + slots - 1) / slots);
    if (cost < best_cost) { best_cost = cost; best_shift = sh; }
  }
  g.shift = best_shift;
  g.lpr = 1 << g.shift;
  g.rpw = 64 >> g.shift;
  return g;
}

template <int VEC>
struct CellVec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ CellVec<VEC> load_cells(const float* p) {
  CellVec<VEC> r;
  if (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1 % VEC] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void store_cells(float* p, const CellVec<VEC>& r) {
  if (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
  } else {
    p[0] = r.v[0];
  }
}
// The last group of a row of a grid that is not a multiple of VEC wide hangs over into the next row (or past the map): those
// cells read as 0 and are never written (another lane owns them).
template <int VEC>
__device__ __forceinline__ CellVec<VEC> load_cells_row(const float* rowp, int y, int gy) {
  if (VEC == 1 || y + VEC <= gy) return load_cells<VEC>(rowp + y);
  CellVec<VEC> r;
#pragma unroll
  for (int q = 0; q < VEC; ++q) r.v[q] = y + q < gy ? rowp[y + q] : 0.f;
  return r;
}
template <int VEC>
__device__ __forceinline__ void store_cells_row(float* rowp, int y, int gy, const CellVec<VEC>& r) {
  if (VEC == 1 || y + VEC <= gy) { store_cells<VEC>(rowp + y, r); return; }
#pragma unroll
  for (int q = 0; q < VEC; ++q)
    if (y + q < gy) rowp[y + q] = r.v[q];
}
// The four flip decisions of a group of cells lin..lin+3 (any alignment) from the Philox words of counter lin >> 2 and, when the
// group straddles it, lin >> 2 + 1: word k of counter c belongs to cell 4 c + k (oracle/ipp_oracle.py::philox_correctness).
__device__ __forceinline__ uint32_t philox_flip_bits4(uint32_t lin, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                     uint32_t thr, bool straddle_possible) {
  const Philox4 a = ippm_philox(lin >> 2, c1, c2, c3, k0, k1);
  uint32_t bits = (a.v[0] < thr ? 1u : 0u) | (a.v[1] < thr ? 2u : 0u) | (a.v[2] < thr ? 4u : 0u) | (a.v[3] < thr ? 8u : 0u);
  if (straddle_possible) {   // (uniform: the grid is not a multiple of 4 wide)
    const Philox4 b = ippm_philox((lin >> 2) + 1u, c1, c2, c3, k0, k1);
    bits |= ((b.v[0] < thr ? 1u : 0u) | (b.v[1] < thr ? 2u : 0u) | (b.v[2] < thr ? 4u : 0u) | (b.v[3] < thr ? 8u : 0u)) << 4;
    bits = (bits >> (lin & 3u)) & 0xFu;
  }
  return bits;
}

// observation bits of a lane's cell group in a code / flips tile: low nibble of one byte (VEC == 4) or one byte per cell
template <int VEC>
__device__ __forceinline__ size_t tile_index(int row, int col, int S) {  // col = y - (yu & ~3)
  return VEC == 4 ? (size_t)row * (S >> 2) + (col >> 2) : (size_t)row * S + col;
}
template <int VEC>
__device__ __forceinline__ uint32_t load_bits(const uint8_t* tile, int row, int col, int S) {
  return tile[tile_index<VEC>(row, col, S)] & (VEC == 4 ? 0xFu : 1u);
}
template <int VEC>
__device__ __forceinline__ void store_bits(uint8_t* tile, int row, int col, int S, uint32_t bits) {
  tile[tile_index<VEC>(row, col, S)] = (uint8_t)bits;
}

// ---- incremental area sums (K6 state) ----------------------------------------------------------------------------
// The network inputs need the exact area average G x G -> 11 x 11 of every belief map at every step
// (utils/state.py:22-41, cv2.resize INTER_AREA).  Output bin b of an axis of n cells covers [b*n/11, (b+1)*n/11); in
// units of 1/11 cell everything is an integer: cell i overlaps bin b0 = floor(11 i / n) by nA = min(11, (b0+1) n - 11 i)
// and bin b0+1 by 11 - nA.  area[map][bx][by] = sum_cells nr(x,bx) nc(y,by) sigmoid(L(x,y)) (float64) is kept up to
// date by the kernels that write maps: each adds nr nc (sigmoid(L_new) - sigmoid(L_old)) for the cells it changes, so
// K6 never streams a map.  The average is area / (gx gy); the all-prior map has area = 0.5 gx gy in every bin.
#define IPPM_AREA_LD 12  // LDS leading dimension: bin index 11 only ever receives zero weight

// exact floor(n / d) for 0 <= n < 2^20, 0 < d <= 2^14 via one float multiply (inv = 1/d within 1 ulp)
__device__ __forceinline__ int area_bin(int n11, float inv_dim) { return (int)(((float)n11 + 0.5f) * inv_dim); }

template <int VEC>
struct AreaCols {   // column-only part, fixed while a lane walks down the rows of its column group
  int cb;           // first column bin of the group (a VEC-cell group spans at most two bins: 4 < n/11 whenever VEC == 4)
  float wA[VEC];    // weight of cell q into bin cb; 11 - wA goes into cb + 1
};
template <int VEC>
__device__ __forceinline__ AreaCols<VEC> area_cols(int y, int gy, float inv_gy) {
  AreaCols<VEC> a;
  a.cb = area_bin(11 * y, inv_gy);
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    const int n = 11 * (y + q);
    const int rem = (a.cb + 1) * gy - n;  // <= 0: the cell starts in bin cb + 1
    a.wA[q] = (float)min(max(rem, 0), 11);
  }
  return a;
}

// (float64: a lane adds up to a dozen rows of weighted sigmoid differences of size ~0.5 x 121 before it flushes, and bins of
// cells near 0 or 1 keep only a small remainder of those sums -- the entropy planes then amplify its error elevenfold)
struct AreaAcc {  // per-lane partial sums for the row bins rb, rb+1 and the column bins cb, cb+1
  double a0A, a0B, a1A, a1B;
  int rb;
  __device__ __forceinline__ void init() { a0A = a0B = a1A = a1B = 0.0; rb = -1; }
  __device__ __forceinline__ void put(double* s_area, int row_bin, int cb, double vA, double vB) {
    if (vA != 0.0) atomicAdd(&s_area[row_bin * IPPM_AREA_LD + cb], vA);      // ds_add_f64
    if (vB != 0.0) atomicAdd(&s_area[row_bin * IPPM_AREA_LD + cb + 1], vB);
  }
  __device__ __forceinline__ void flush(double* s_area, int cb) {
    if (rb >= 0) { put(s_area, rb, cb, a0A, a0B); put(s_area, rb + 1, cb, a1A, a1B); }
    init();
  }
  // rows arrive in ascending order; cA / cB = sum_q wA[q] d[q], sum_q (11 - wA[q]) d[q] of one row
  __device__ __forceinline__ void add(double* s_area, int cb, int x, int gx, float inv_gx, float cA, float cB) {
    const int n = 11 * x;
    const int rbn = area_bin(n, inv_gx);
    if (rbn != rb) {
      if (rb >= 0) {
        put(s_area, rb, cb, a0A, a0B);
        if (rbn == rb + 1) { a0A = a1A; a0B = a1B; }
        else { put(s_area, rb + 1, cb, a1A, a1B); a0A = a0B = 0.0; }
      }
      a1A = a1B = 0.0;
      rb = rbn;
    }
    const float nA = (float)min((rbn + 1) * gx - n, 11), nB = 11.f - nA;
    a0A += (double)(nA * cA); a0B += (double)(nA * cB); a1A += (double)(nB * cA); a1B += (double)(nB * cB);
  }
};

// sigmoid(a) - sigmoid(b) = (e_b - e_a) / ((1 + e_a)(1 + e_b)), e = exp(-L): three transcendentals instead of four, exact
// zero for a == b.  |L| is capped at 40 (sigmoid is 0 / 1 to float32 precision from |L| = 17 on) so that the product of the
// two denominators stays finite: e^40 e^40 = 5.5e34.  Infinite log-odds do occur: the reference's sensor model is noise-free
// at altitudes other than 5 / 10 / 15 m (sensor_models.py:13-22), a measurement there sets a cell to exactly 0 or 1.
__device__ __forceinline__ float sigmoid_diff(float a, float b) {
  const float ea = __expf(-fminf(fmaxf(a, -40.f), 40.f)), eb = __expf(-fminf(fmaxf(b, -40.f), 40.f));
  return (eb - ea) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
}
// (Two separate sigmoids would carry an absolute error of ~6e-8 each; cells saturated at the clip all hold the same value and
// make the same transitions, so those errors add up coherently over a bin -- 2e-7 on an area average near 1, which the entropy
// plane amplifies elevenfold.  The quotient form's error is relative to the difference.)

__device__ __forceinline__ void area_lds_clear(double* s_area) {
  for (int q = threadIdx.x; q < IPPM_FEAT * IPPM_AREA_LD + IPPM_AREA_LD; q += blockDim.x) s_area[q] = 0.0;
}
// one float64 atomic per touched bin and workgroup
__device__ __forceinline__ void area_lds_commit(const double* s_area, double* area_map) {
  for (int q = threadIdx.x; q < IPPM_FEAT * IPPM_FEAT; q += blockDim.x) {
    const double v = s_area[(q / IPPM_FEAT) * IPPM_AREA_LD + q % IPPM_FEAT];
    if (v != 0.0) atomicAdd(&area_map[q], v);
  }
}

// contribution of one row of a lane's cell group: d[q] = sigmoid(new) - sigmoid(old) (0 for unchanged cells)
template <int VEC>
__device__ __forceinline__ void area_row(AreaAcc& acc, double* s_area, const AreaCols<VEC>& ac, int x, int gx, float inv_gx,
                                         const float* d) {
  float sd = 0.f, cA = 0.f;
#pragma unroll
  for (int q = 0; q < VEC; ++q) { sd += d[q]; cA += ac.wA[q] * d[q]; }
  acc.add(s_area, ac.cb, x, gx, inv_gx, cA, 11.f * sd - cA);
}

#endif  // __HIPCC__
