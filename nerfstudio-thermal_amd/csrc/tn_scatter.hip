// Trilinear scatter-add of d(encoding) into a hash table's gradient (+ d position), shared by the proposal grids and the main grid.
// Three paths, newest last in the file (TN_SCATTER_MODE): 0 = global float atomics with dense replicas (round 1; below), 1 = "binned": records
// reserved into per-bucket global arrays, summed per bucket in LDS (rounds 2-5), 2 = "segmented" (round 5, the default): records in regions that
// belong to one bin block, no global atomic at all, a fold that gathers its bucket's segment from every bin block.
//
// Global float atomics on MI355X are bound by 64-byte REQUESTS (~21 G requests/s, scripts/microbench/atomic_shapes.hip): lanes of one
// wave-instruction that fall into the same 64-B line cost one request.  The kernel is laid out for that:
//   * 4 lanes per sample: lane q = (x-corner choice, feature) = [f.x, f.y, c.x, c.y].  The floor and ceil corners differ by 1 in x, and
//     the reference's hash (x*1 ^ y*P1 ^ z*P2) & mask keeps x in the low bits, so 7 times out of 8 the two entries (2 x float2 = 16 B or
//     within one 64-B line) are served by ONE request: a sample costs ~4.5 requests per level instead of 16.
//     The four lanes must be ADJACENT (one quad): the atomic path works quad by quad, and with one lane in each 16-lane row (which would let
//     the scans below run as DPP row shifts) a run tail activates four quads instead of one -- measured 20 % slower overall even though
//     the all-lanes-active microbenchmark (scripts/microbench/atomic_lane_map.hip) sees no difference.
//   * work items in patch order (tn_patch_order) and run-length merging across consecutive samples of the wave: samples in the same
//     grid cell are summed in registers (stride-4 segmented scan) and only the run's last sample issues the atomics.
#include "tn_common.h"
#include "tn_field_dpos.h"
#include "tn_pose_finish.h"
#include <stdlib.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ float seg_sum4(float v, int start, int lane) {
  int sl = lane >> 2;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    float t = __shfl_up(v, 4 * o, 64);
    if (sl - o >= start) v += t;
  }
  return v;
}

__global__ void __launch_bounds__(256) k_grid_scatter(GridK g, const float* __restrict__ origins, const float* __restrict__ directions,
                                                      const float* __restrict__ e_bins, const float* __restrict__ g_enc, int ld, int64_t N, int S,
                                                      float* __restrict__ d_origins, float* __restrict__ d_directions, int level_groups, ReplicaK rk) {
  const bool want_dpos = d_origins != nullptr;
  const int lane = tn_lane();
  const int q = lane & 3, xc = q >> 1, ft = q & 1;
  const int64_t P = N * (int64_t)S;
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 2);
  const int64_t iters = tn_cdiv(P, stride);
  for (int64_t it = 0; it < iters; ++it) {
    int64_t i = it * stride + (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const bool live = i < P;
    if (!live) i = P - 1;
    int64_t ray;
    int s;
    tn_patch_order(i, N, S, ray, s);
    const int64_t p = ray * S + s;
    const float* o = origins + ray * 3;
    const float* d = directions + ray * 3;
    const float* eb = e_bins + ray * (S + 1) + s;
    const float st = eb[0], en = eb[1];
    const Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], st, en);
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;
    // blockIdx.y = level group: levels l == group (mod level_groups).  Splitting the levels gives the dispatcher more, shorter work items
    // (the 16-level main grid at 4096 rays is otherwise 1.5 waves of resident blocks: a 25% tail).
    // Two passes over the levels: (1) every LOAD (the level's gradient and, for d position, the corner values), (2) every ATOMIC.  vmcnt
    // retires in order, so a load issued behind an atomic waits for it (~3000 cycles when every CU is adding): keeping the loads first lets the
    // wave fire all its atomics back to back.
    float gvs[TN_MAX_LEVELS];
#pragma unroll
    for (int li = 0; li < TN_MAX_LEVELS; ++li) {
      int l = blockIdx.y + li * level_groups;
      gvs[li] = (live && l < g.L) ? g_enc[ld > 0 ? p * ld + 2 * l + ft : (int64_t)l * 2 * P + 2 * p + ft] : 0.0f;
    }
    if (want_dpos) {
#pragma unroll 1
      for (int li = 0; li < TN_MAX_LEVELS; ++li) {
        int l = blockIdx.y + li * level_groups;
        if (l >= g.L) break;
        float gv = gvs[li];
        const float res = g.res[l];
        const uint32_t level_off = (uint32_t)l * g.tsize;
        float sx = c.px * res, sy = c.py * res, sz = c.pz * res;
        float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
        uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf, fz = (uint32_t)(int)fzf;
        uint32_t cx = (uint32_t)(int)ceilf(sx), cy = (uint32_t)(int)ceilf(sy), cz = (uint32_t)(int)ceilf(sz);
        float ox = sx - fxf, oy = sy - fyf, oz = sz - fzf;
        float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
        uint32_t xi = xc ? cx : fx;
        float wxv = xc ? ox : ux;
        uint32_t hcy = cy * TN_PRIME_Y, hfy = fy * TN_PRIME_Y, hcz = cz * TN_PRIME_Z, hfz = fz * TN_PRIME_Z;
        const uint32_t hy[4] = {hcy, hfy, hcy, hfy};
        const uint32_t hz[4] = {hcz, hcz, hfz, hfz};
        const float wy[4] = {oy, uy, oy, uy};
        const float wz[4] = {oz, oz, uz, uz};
        float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uint32_t idx = ((xi ^ hy[k] ^ hz[k]) & g.mask) + level_off;
          float2 t = g.table[idx];
          float tv = ft ? t.y : t.x;
          a += wy[k] * wz[k] * tv;
          b += ((k & 1) ? -1.0f : 1.0f) * wz[k] * tv;
          cc += wy[k] * ((k & 2) ? -1.0f : 1.0f) * tv;
        }
        dpx += (xc ? 1.0f : -1.0f) * a * gv * res;
        dpy += wxv * b * gv * res;
        dpz += wxv * cc * gv * res;
      }
    }
#pragma unroll 1
    for (int li = 0; li < TN_MAX_LEVELS; ++li) {  // wave-uniform trip count
      int l = blockIdx.y + li * level_groups;
      if (l >= g.L) break;
      float gv = gvs[li];
      const float res = g.res[l];
      const uint32_t level_off = (uint32_t)l * g.tsize;
      float sx = c.px * res, sy = c.py * res, sz = c.pz * res;
      float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
      uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf, fz = (uint32_t)(int)fzf;
      uint32_t cx = (uint32_t)(int)ceilf(sx), cy = (uint32_t)(int)ceilf(sy), cz = (uint32_t)(int)ceilf(sz);
      float ox = sx - fxf, oy = sy - fyf, oz = sz - fzf;
      float ux = 1.0f - ox, uy = 1.0f - oy, uz = 1.0f - oz;
      // run key over consecutive samples (lanes 4 apart): floor cell + which axes sit exactly on the lattice
      uint32_t k1 = fx | (fy << 16);
      uint32_t k2 = fz | ((ox == 0.0f) ? 1u << 16 : 0u) | ((oy == 0.0f) ? 1u << 17 : 0u) | ((oz == 0.0f) ? 1u << 18 : 0u);
      uint32_t p1 = __shfl_up(k1, 4, 64), p2 = __shfl_up(k2, 4, 64);
      bool head = (lane < 4) || (k1 != p1) || (k2 != p2);
      // run start / run tail from ONE ballot instead of a 4-step shuffle scan plus a shuffle: the four lanes of a sample agree on `head`, so
      // the highest set bit at or below this lane lies in the head sample of its run (the kernel is bound by the latency of these dependent
      // cross-lane chains: TCC_EA0_ATOMIC_LEVEL shows the atomic pipe under-filled on the proposal grids, not over-subscribed)
      const unsigned long long H = __ballot(head);
      const int start = (63 - __clzll(H & (~0ull >> (63 - lane)))) >> 2;  // in sample units
      const bool tail = (lane >= 60) || ((H >> (lane + 4)) & 1ull);
      // this lane's x corner and its 4 (y,z) partners: (c,c) (f,c) (c,f) (f,f)
      uint32_t xi = xc ? cx : fx;
      float wxv = xc ? ox : ux;
      uint32_t hcy = cy * TN_PRIME_Y, hfy = fy * TN_PRIME_Y, hcz = cz * TN_PRIME_Z, hfz = fz * TN_PRIME_Z;
      const float wy[4] = {oy, uy, oy, uy};
      const float wz[4] = {oz, oz, uz, uz};
      // wave-uniform: the hashed gradient itself, or this wave's private dense replica of the level
      const uint32_t kind = (rk.kinds >> (2 * l)) & 3u;
      const bool dense = kind == TN_REP_DENSE;
      float2* base = g.grad + level_off;
      if (kind != TN_REP_NONE) base = rk.rep + rk.off[l] + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % (unsigned)rk.R[l]) * rk.n[l];  // replica per WAVE: the waves of a block walk adjacent samples
      if (dense) {
        const uint32_t r1 = (uint32_t)(int)ceilf(res) + 1u, top = r1 - 1u;
        xi = min(xi, top);  // positions are in [0,1] by construction; the clamp only keeps a corrupt input inside the replica
        hcy = min(cy, top) * r1; hfy = min(fy, top) * r1;
        hcz = min(cz, top) * r1 * r1; hfz = min(fz, top) * r1 * r1;
      }
      const uint32_t hy[4] = {hcy, hfy, hcy, hfy};
      const uint32_t hz[4] = {hcz, hcz, hfz, hfz};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t idx = dense ? (xi + hy[k] + hz[k]) : ((xi ^ hy[k] ^ hz[k]) & g.mask);
        float w = wxv * wy[k] * wz[k];
        float v = seg_sum4(w * gv, start, lane);
        if (tail && v != 0.0f) {
          unsafeAtomicAdd(reinterpret_cast<float*>(base + idx) + ft, v);
          if (g.nonfinite != nullptr && (v - v) != 0.0f) *g.nonfinite = 1.0f;
        }
      }
    }
    if (want_dpos) {
      // sum the 4 lanes of the sample, then reduce the wave's 16 samples per ray
      dpx += __shfl_xor(dpx, 1, 64); dpx += __shfl_xor(dpx, 2, 64);
      dpy += __shfl_xor(dpy, 1, 64); dpy += __shfl_xor(dpy, 2, 64);
      dpz += __shfl_xor(dpz, 1, 64); dpz += __shfl_xor(dpz, 2, 64);
      float wx, wy_, wz_;
      tn_contract_bwd(c, dpx, dpy, dpz, wx, wy_, wz_);
      if (!live) { wx = wy_ = wz_ = 0.0f; }
      float tm = (st + en) / 2.0f;
      float v[6] = {wx, wy_, wz_, wx * tm, wy_ * tm, wz_ * tm};
      // patch order: sample (lane>>2) of the wave belongs to ray class (lane>>2)&3 -> lanes 16 and 32 apart share a ray
      int r32 = (int)ray;
      int lead = __shfl(r32, lane & 15, 64);
      if (__all(r32 == lead)) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float r = v[k];
          r += __shfl_xor(r, 16, 64);
          r += __shfl_xor(r, 32, 64);
          if (lane < 16 && q == 0 && r != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), r);
        }
      } else if (q == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (v[k] != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), v[k]);
      }
    }
  }
}

// Folds the replicas into the hashed gradient: thread = one entry of one replicated level.
__global__ void __launch_bounds__(256) k_replica_reduce(GridK g, ReplicaK rk, float2* __restrict__ dense_out) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rk.total) return;
  int l = 0;
#pragma unroll 1
  for (int i = 0; i < g.L; ++i)
    if (((rk.kinds >> (2 * i)) & 3u) != TN_REP_NONE && rk.first[i] <= e) l = i;  // first[] grows with the level
  const uint32_t t = e - rk.first[l], n = rk.n[l];
  const float2* src = rk.rep + rk.off[l] + t;
  float sx = 0.f, sy = 0.f;
  const int R = rk.R[l];
  int r = 0;
  for (; r + 4 <= R; r += 4) {  // four independent loads in flight (the kernel is one short latency chain per thread otherwise)
    float2 v0 = src[(size_t)r * n], v1 = src[(size_t)(r + 1) * n], v2 = src[(size_t)(r + 2) * n], v3 = src[(size_t)(r + 3) * n];
    sx += (v0.x + v1.x) + (v2.x + v3.x);
    sy += (v0.y + v1.y) + (v2.y + v3.y);
  }
  for (; r < R; ++r) {
    float2 v = src[(size_t)r * n];
    sx += v.x;
    sy += v.y;
  }
  if (dense_out != nullptr) {  // data-parallel exchange of the coarse levels: hand the per-cell sums over un-hashed (tn_field_dense_fold follows)
    dense_out[e] = make_float2(sx, sy);
    return;
  }
  if (sx == 0.0f && sy == 0.0f) return;  // untouched entries keep an exactly-zero gradient (Adam's eps = 1e-15 makes that matter)
  const uint32_t r1 = (uint32_t)(int)ceilf(g.res[l]) + 1u;
  uint32_t x = t % r1, y = (t / r1) % r1, z = t / (r1 * r1);
  uint32_t idx = ((x ^ (y * TN_PRIME_Y) ^ (z * TN_PRIME_Z)) & g.mask) + (uint32_t)l * g.tsize;
  float* dst = reinterpret_cast<float*>(g.grad + idx);  // several dense entries may share a slot
  if (sx != 0.0f) unsafeAtomicAdd(dst, sx);
  if (sy != 0.0f) unsafeAtomicAdd(dst + 1, sy);
  if (g.nonfinite != nullptr && ((sx - sx) + (sy - sy)) != 0.0f) *g.nonfinite = 1.0f;
}

// ======================================================================================================================================
// Atomic-free ("binned") path -- TN_SCATTER_MODE=1; the segmented path further down is the default and shares its record format, bucket size and
// LDS-image fold.  Global float atomics execute at the memory side at ~20 G 64-byte requests/s whatever the
// kernel does (MI355X_MICROARCH.md, Global float atomics), plain coalesced stores run 4-5x faster, so every contribution is WRITTEN once
// and summed on chip:
//   pass 1  k_grid_bin : lane = sample (patch order).  Per level: 8 corner slots + values; runs of consecutive samples in one grid cell are
//           summed in registers on the coarser levels (segmented wave scan); the surviving (slot, value) records are ranked per bucket
//           (bucket = 2^TN_BIN_SLICE_LOG2 consecutive slots of one level) with LDS counters, staged in LDS in bucket order, and copied out
//           as contiguous runs into the buckets' global arrays (one returning integer atomic per block, level and bucket reserves the run).
//   pass 2  k_grid_fold: block = (bucket, chunk of its records): sums the records into an LDS image of the bucket's slots (double-precision LDS
//           atomic adds, see below), then adds the non-zero slots into the table gradient with coalesced plain read-modify-writes
//           (atomics only when a bucket is split over several chunks).
// A bucket that overflows its capacity (never on hashed levels with anything like a real batch) falls back to global atomics for the excess,
// so the result is right for any input.  The sample position's gradient comes out of pass 1 (same corner gathers as the forward).
// What was measured on the way here (lane-serialised LDS float atomics, microsecond barrier intervals, XCD-skewed block order):
// profiles/r02_scatter_alternatives.md.
struct BinK {
  uint16_t* idx;     // [L][nslices][cap] slot inside the bucket
  float2* val;       // same shape
  uint32_t* count;   // [L][nslices] x cstride words, zeroed before pass 1
  uint32_t cstride;  // words between two counters (1; one 64-B line per counter (16) was measured 15 % SLOWER on the main grid: more lines to fetch, no gain)
  uint32_t cap;      // records per bucket (multiple of 8)
  uint32_t level_stride;  // records between levels = nslices * cap
  int slice_log2;    // log2 slots per bucket
  int nslices;       // buckets per level (power of two, <= TN_BIN_MAX_SLICES)
  uint32_t merge_mask;  // bit l: sum runs of same-cell neighbours at level l before writing
  // fold work items: a bucket's records are cut into chunks of chunk[l] records, one block each; level l owns blocks [blk0[l], blk0[l+1]).
  // Coarse levels have few live slots per bucket and hundreds of records per slot (same-address LDS atomics execute one after the other):
  // their buckets are cut into smaller chunks that spread over more CUs.
  uint32_t chunk[TN_MAX_LEVELS];
  uint32_t blk0[TN_MAX_LEVELS + 1];
  unsigned long long* trace;  // TN_FOLD_TRACE diagnostics: per fold block {start, after load+zero, after passes, end} wall-clock stamps, or NULL
};

#ifndef BIN_THREADS
#define BIN_THREADS 512  // 256: 198 us for the main grid, 512: 172 us (half the reservations, runs twice as long); 384 / 640 / 768 / 1024: 190 / 204 / 192 / 177 us
#endif
#ifndef BIN_MAX_COUNTERS
#define BIN_MAX_COUNTERS 1024  // (levels handled by one block) x (buckets per level)
#endif

// corner slots of one level + run structure of the wave (which lanes write: the last lane of every run of same-cell samples)
struct BinLevel {
  uint32_t idx[8];            // table slots of the 8 corners (inside the level)
  float wx[2], wy[2], wz[2];  // [0] floor corner, [1] ceil corner
  bool emit;
  int start;    // first lane of this lane's run (merge levels)
  int maxlen;   // power of two >= the longest run of the wave, capped at 16 (merge levels): which steps of the scan are needed
  bool cross;   // some run continues over a boundary between rows of 16 lanes (merge levels)
};
__device__ __forceinline__ void bin_level(const Contracted& c, float res, uint32_t mask, bool merge, bool live, int lane, BinLevel& b) {
  const float sx = c.px * res, sy = c.py * res, sz = c.pz * res;
  const float fxf = floorf(sx), fyf = floorf(sy), fzf = floorf(sz);
  const uint32_t fx = (uint32_t)(int)fxf, fy = (uint32_t)(int)fyf, fz = (uint32_t)(int)fzf;
  const uint32_t cx = (uint32_t)(int)ceilf(sx), cy = (uint32_t)(int)ceilf(sy), cz = (uint32_t)(int)ceilf(sz);
  const float ox = sx - fxf, oy = sy - fyf, oz = sz - fzf;
  b.wx[0] = 1.0f - ox; b.wx[1] = ox;
  b.wy[0] = 1.0f - oy; b.wy[1] = oy;
  b.wz[0] = 1.0f - oz; b.wz[1] = oz;
  const uint32_t hx[2] = {fx, cx}, hy[2] = {fy * TN_PRIME_Y, cy * TN_PRIME_Y}, hz[2] = {fz * TN_PRIME_Z, cz * TN_PRIME_Z};
#pragma unroll
  for (int k = 0; k < 8; ++k) b.idx[k] = (hx[k & 1] ^ hy[(k >> 1) & 1] ^ hz[k >> 2]) & mask;  // k = (z, y, x) corner bits
  b.emit = live;
  b.start = lane;
  b.maxlen = 1;
  b.cross = false;
  if (merge) {
    // runs of consecutive lanes in one cell (same floor corner, same on-lattice flags => same 8 slots)
    const uint32_t k1 = fx | (fy << 16);
    const uint32_t k2 = fz | ((ox == 0.0f) ? 1u << 16 : 0u) | ((oy == 0.0f) ? 1u << 17 : 0u) | ((oz == 0.0f) ? 1u << 18 : 0u);
    // the lane before: DPP wave_shr:1 (vector ALU; lane 0 keeps its own key, it is a head anyway)
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_update_dpp((int)k1, (int)k1, 0x138, 0xf, 0xf, false);
    const uint32_t p2 = (uint32_t)__builtin_amdgcn_update_dpp((int)k2, (int)k2, 0x138, 0xf, 0xf, false);
    const bool head = (lane == 0) || (k1 != p1) || (k2 != p2) || !live;
    const unsigned long long H = __ballot(head);
    b.start = 63 - __clzll(H & (~0ull >> (63 - lane)));
    const bool tail = (lane == 63) || ((H >> (lane + 1)) & 1ull);
    // which scan steps any run of the wave needs (wave-uniform, from ballots: no cross-lane reduction)
    const int back = lane - b.start;
    b.maxlen = __ballot(back >= 8) ? 16 : (__ballot(back >= 4) ? 8 : (__ballot(back >= 2) ? 4 : (__ballot(back >= 1) ? 2 : 1)));
    b.cross = (~H & 0x0001000100010000ull) != 0ull;
    b.emit = live && tail;
  }
}
// Sums of the 16 values over every run of same-cell lanes, delivered to the run's LAST lane: a segmented inclusive scan on the vector ALU
// (DPP), not through ds_bpermute -- the 96 permutes per level of the shuffle version kept the CU's one LDS pipe busy for 47 us of the level-0
// proposal grid's 164-us bin pass.  Rows of 16 lanes first (row_shr 1, 2, 4, 8; a source outside the row reads 0), then the carries across
// the row boundaries as in the classic wave scan (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3), each addition only where
// the lane's run reaches that far back.
template <int CTRL>
__device__ __forceinline__ float bin_dpp(float v) {  // every row enabled, invalid source lanes read 0: folds into the consumer (v_fmac_f32_dpp)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ void bin_scan_step(float (&vx)[8], float (&vy)[8], bool take) {
  const float m = take ? 1.0f : 0.0f;  // v += m * shifted(v): one instruction per value (the sums stay exact: m is 0 or 1)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    vx[k] = __builtin_fmaf(bin_dpp<CTRL>(vx[k]), m, vx[k]);
    vy[k] = __builtin_fmaf(bin_dpp<CTRL>(vy[k]), m, vy[k]);
  }
}
__device__ __forceinline__ void bin_run_sums(float (&vx)[8], float (&vy)[8], int lane, const BinLevel& b) {
  const int back = lane - b.start;  // lanes of the run before this one
  bin_scan_step<0x111>(vx, vy, back >= 1);
  if (b.maxlen > 2) bin_scan_step<0x112>(vx, vy, back >= 2);
  if (b.maxlen > 4) bin_scan_step<0x114>(vx, vy, back >= 4);
  if (b.maxlen > 8) bin_scan_step<0x118>(vx, vy, back >= 8);
  if (b.cross) {  // some run continues over a row boundary (wave-uniform)
    bin_scan_step<0x142>(vx, vy, (lane & 16) && b.start < (lane & 48));  // rows 1, 3 += lane 15 / 47 (sum of the run's part in the row before)
    bin_scan_step<0x143>(vx, vy, (lane & 32) && b.start < 32);            // rows 2, 3 += lane 31 (everything of the run in rows 0-1)
  }
}

template <bool WANT_DPOS>
__global__ void __launch_bounds__(BIN_THREADS, 4) k_grid_bin(GridK g, const float* __restrict__ origins, const float* __restrict__ directions,
                                                          const float* __restrict__ e_bins, const float* __restrict__ g_enc, int ld, int64_t N, int S,
                                                          float* __restrict__ d_origins, float* __restrict__ d_directions, int level_groups, BinK bk) {
  // s_cnt: phase A: records per (level, bucket); phase B: next free staging position of the bucket (starts at the bucket's offset s_loff)
  // s_dl: per (level, bucket) {delta, lim}: staged record j of the bucket goes to position j + delta of the level's global record array if
  //       j < lim (delta = bucket * cap + reserved base - staging offset; lim = staging offset + room left in the bucket), else the bucket is full
  __shared__ uint32_t s_cnt[BIN_MAX_COUNTERS], s_loff[BIN_MAX_COUNTERS];
  __shared__ uint2 s_dl[BIN_MAX_COUNTERS];
  __shared__ uint32_t s_tot[TN_MAX_LEVELS], s_lb[TN_MAX_LEVELS];
  __shared__ uint32_t s_idx[BIN_THREADS * 8];
  __shared__ float2 s_val[BIN_THREADS * 8];
  const int lane = tn_lane();
  const int tid = threadIdx.x;
  const int ns = bk.nslices;
  const int64_t P = N * (int64_t)S;
  int64_t i = (int64_t)blockIdx.x * BIN_THREADS + tid;
  const bool live = i < P;
  if (!live) i = P - 1;
  int64_t ray;
  int s;
  tn_patch_order(i, N, S, ray, s);
  const int64_t p = ray * S + s;
  const float* o = origins + ray * 3;
  const float* d = directions + ray * 3;
  const float* eb = e_bins + ray * (S + 1) + s;
  const float st = eb[0], en = eb[1];
  const Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], st, en);
  const int nlev = (g.L - (int)blockIdx.y + level_groups - 1) / level_groups;  // levels of this block: blockIdx.y + li * level_groups
  // The block is a chain of dependent latencies (a global load per level, the returning reservation atomics, ten barriers), not a
  // throughput problem: 65 % of its wave-cycles were spent parked (profiles/r03_pmc_summary.md).  So every load whose address is known
  // up front is issued up front -- d enc of ALL the block's levels here -- and the reservations below are waited for only after the first
  // level has been staged.
  // (Not in the d-position variant: at four waves per SIMD it has 128 registers and spills with these prefetches -- measured 128 -> 131 us.)
  constexpr int GVP = WANT_DPOS ? 0 : 5;
  float2 gvp[GVP + 1];
#pragma unroll
  for (int li = 0; li < GVP; ++li) {
    const int l = blockIdx.y + li * level_groups;
    gvp[li] = (live && li < nlev) ? *reinterpret_cast<const float2*>(ld > 0 ? g_enc + p * ld + 2 * l : g_enc + (int64_t)l * 2 * P + 2 * p) : make_float2(0.f, 0.f);
  }
  // ---- phase A: how many records this block sends to every bucket, one reservation per bucket for all levels at once (a returning global
  // atomic takes microseconds: inside the level loop it was the critical path)
  for (int t = tid; t < nlev * ns; t += BIN_THREADS) s_cnt[t] = 0;
  __syncthreads();
  BinLevel bfirst;  // the first level's corners and run structure are kept for phase B (one of the nlev recomputations less)
#pragma unroll 1
  for (int li = 0; li < nlev; ++li) {
    const int l = blockIdx.y + li * level_groups;
    BinLevel b;
    bin_level(c, g.res[l], g.mask, (bk.merge_mask >> l) & 1u, live, lane, b);
    if (!WANT_DPOS && li == 0) bfirst = b;  // (the d-position variant has no registers to spare: 128 at four waves per SIMD)
    if (b.emit) {
#pragma unroll
      for (int k = 0; k < 8; ++k) atomicAdd(&s_cnt[li * ns + (b.idx[k] >> bk.slice_log2)], 1u);
    }
  }
  __syncthreads();
  // one wave per level: offsets of the buckets inside the staging area, and one returning global atomic per non-empty bucket reserves the run
  // in the bucket's global array.  The atomics' results (gb) are NOT waited for here: they go to s_gbase after the first level's staging.
  constexpr int GB = TN_BIN_MAX_SLICES / 64;
  uint32_t gb[GB];
  const int my_li = tid >> 6;  // the level this wave reserves for (if < nlev); further levels of blocks with more than 8 levels: the loop below
#pragma unroll
  for (int u = 0; u < GB; ++u) gb[u] = 0u;
  for (int li = my_li; li < nlev; li += BIN_THREADS / 64) {
    const int l = blockIdx.y + li * level_groups;
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < GB; ++u) {
      const int base = u * 64;
      if (base >= ns) break;
      const int sl = base + lane;
      const uint32_t cnt = sl < ns ? s_cnt[li * ns + sl] : 0u;
      uint32_t inc = cnt;
#pragma unroll
      for (int o2 = 1; o2 < 64; o2 <<= 1) {
        const uint32_t t = __shfl_up(inc, o2, 64);
        if (lane >= o2) inc += t;
      }
      uint32_t r = 0u;
      if (sl < ns) {
        const uint32_t loff = run + inc - cnt;
        s_loff[li * ns + sl] = loff;
        r = cnt ? atomicAdd(&bk.count[((size_t)l * ns + sl) * bk.cstride], cnt) : 0u;
        s_cnt[li * ns + sl] = loff;  // becomes the bucket's staging cursor of phase B
        if (li != my_li) s_dl[li * ns + sl] = make_uint2((uint32_t)sl * bk.cap + r - loff, loff + (r < bk.cap ? bk.cap - r : 0u));
      }
      if (li == my_li) gb[u] = r;  // (first level of this wave: deferred)
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) s_tot[li] = run;
  }
  __syncthreads();
  // ---- phase B: values, ranks, staging in bucket order, coalesced copy-out -- in ROUNDS of consecutive levels of the block.  Phase A knows how
  // many records every level stages (s_tot), so as many levels as fit the staging area share one round = one pair of block barriers: on the
  // levels where same-cell runs are merged a block stages a few hundred records, not 4096 (the proposal grids: all five levels in one round),
  // and the pass spent more than half of its wave-cycles parked at those barriers (profiles/r04_pmc.json).  A staged record carries its level
  // (bits 24+ of s_idx; slots have at most 20 bits on this path) and s_lb holds the level's first staging position of the round.
  float dpx = 0.f, dpy = 0.f, dpz = 0.f;
  const uint32_t smask = (1u << bk.slice_log2) - 1u;
  int li0 = 0;
#pragma unroll 1
  while (li0 < nlev) {
    uint32_t round_total = s_tot[li0];
    int li1 = li0 + 1;
    while (li1 < nlev && round_total + s_tot[li1] <= (uint32_t)(BIN_THREADS * 8)) round_total += s_tot[li1++];
    uint32_t lbase = 0;
#pragma unroll 1
    for (int li = li0; li < li1; ++li) {
      const int l = blockIdx.y + li * level_groups;
      const float res = g.res[l];
      const bool merge = (bk.merge_mask >> l) & 1u;
      BinLevel b;
      if (!WANT_DPOS && li == 0) b = bfirst;
      else bin_level(c, res, g.mask, merge, live, lane, b);
      // row-major [P][ld], or level-major [L][P] float2 (TN_LD_LEVEL_MAJOR: the 64 lanes of a wave then read four runs of 16 consecutive float2
      // instead of 64 pieces of 8 bytes 128 B apart -- 114 MB fetched for the main grid's 25 MB of d enc)
      float2 gv = make_float2(0.f, 0.f);
      if (li < GVP) {
#pragma unroll
        for (int q = 0; q < GVP; ++q)
          if (q == li) gv = gvp[q];
      } else if (live) {
        gv = *reinterpret_cast<const float2*>(ld > 0 ? g_enc + p * ld + 2 * l : g_enc + (int64_t)l * 2 * P + 2 * p);
      }
      if (WANT_DPOS) {
        // d enc / d position from the corner values: s_k = <g, table[corner k]>, then the three one-sided differences of the trilinear form
        const float2* tb = g.table + (size_t)l * g.tsize;
        float sk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float2 t = tb[b.idx[k]];
          sk[k] = gv.x * t.x + gv.y * t.y;
        }
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            ax += b.wy[u] * b.wz[v] * (sk[1 + 2 * u + 4 * v] - sk[0 + 2 * u + 4 * v]);
            ay += b.wx[u] * b.wz[v] * (sk[u + 2 + 4 * v] - sk[u + 0 + 4 * v]);
            az += b.wx[u] * b.wy[v] * (sk[u + 2 * v + 4] - sk[u + 2 * v + 0]);
          }
        dpx += ax * res;
        dpy += ay * res;
        dpz += az * res;
      }
      float vx[8], vy[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = b.wx[k & 1] * b.wy[(k >> 1) & 1] * b.wz[k >> 2];
        vx[k] = w * gv.x;
        vy[k] = w * gv.y;
      }
      if (merge && b.maxlen > 1) bin_run_sums(vx, vy, lane, b);  // the run's last lane ends up with the run's sums
      if (tid == 0) s_lb[li] = lbase;
      if (b.emit) {
        const uint32_t tag = (uint32_t)li << 24;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const uint32_t pos = atomicAdd(&s_cnt[li * ns + (b.idx[k] >> bk.slice_log2)], 1u) + lbase;
          s_idx[pos] = b.idx[k] | tag;
          s_val[pos] = make_float2(vx[k], vy[k]);
        }
      }
      lbase += s_tot[li];
      if (li == 0 && my_li < nlev) {  // the reservations of this wave's level have had the whole first level's staging to come back
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int sl = u * 64 + lane;
          if (u * 64 < ns && sl < ns) {
            const uint32_t loff = s_loff[my_li * ns + sl], r = gb[u];
            s_dl[my_li * ns + sl] = make_uint2((uint32_t)sl * bk.cap + r - loff, loff + (r < bk.cap ? bk.cap - r : 0u));
          }
        }
      }
    }
    __syncthreads();
    if (li1 - li0 == 1) {
      // a round of ONE level (the main grid's fine levels: nothing merges, a level fills the staging area): wave-uniform array bases and 32-bit
      // record offsets -- ~10 vector instructions per record less than the general loop below, on the levels that stage the most records
      const int li = li0;
      const int l = blockIdx.y + li * level_groups;
      uint16_t* __restrict__ idx_l = bk.idx + (size_t)l * bk.level_stride;
      float2* __restrict__ val_l = bk.val + (size_t)l * bk.level_stride;
      for (uint32_t j = tid; j < round_total; j += BIN_THREADS) {
        const uint32_t id = s_idx[j] & 0xffffffu;
        const float2 v = s_val[j];
        const uint2 dl = s_dl[li * ns + (id >> bk.slice_log2)];
        if (j < dl.y) {
          const uint32_t at = j + dl.x;
          idx_l[at] = (uint16_t)(id & smask);
          val_l[at] = v;
        } else {  // bucket full: add straight into the gradient
          float* dst = reinterpret_cast<float*>(g.grad + (size_t)l * g.tsize + id);
          if (v.x != 0.0f) unsafeAtomicAdd(dst, v.x);
          if (v.y != 0.0f) unsafeAtomicAdd(dst + 1, v.y);
          if (g.nonfinite != nullptr && ((v.x - v.x) + (v.y - v.y)) != 0.0f) *g.nonfinite = 1.0f;
        }
      }
    } else
    for (uint32_t J = tid; J < round_total; J += BIN_THREADS) {
      const uint32_t tagged = s_idx[J];
      const float2 v = s_val[J];
      const uint32_t lr = tagged >> 24, id = tagged & 0xffffffu;
      const uint32_t l = blockIdx.y + lr * (uint32_t)level_groups;
      const uint32_t j = J - s_lb[lr];  // position among the level's staged records
      const uint2 dl = s_dl[lr * ns + (id >> bk.slice_log2)];
      if (j < dl.y) {
        const size_t at = (size_t)l * bk.level_stride + (j + dl.x);  // nslices * cap < 2^32 per level
        bk.idx[at] = (uint16_t)(id & smask);
        bk.val[at] = v;
      } else {  // bucket full: add straight into the gradient
        float* dst = reinterpret_cast<float*>(g.grad + (size_t)l * g.tsize + id);
        if (v.x != 0.0f) unsafeAtomicAdd(dst, v.x);
        if (v.y != 0.0f) unsafeAtomicAdd(dst + 1, v.y);
        if (g.nonfinite != nullptr && ((v.x - v.x) + (v.y - v.y)) != 0.0f) *g.nonfinite = 1.0f;
      }
    }
    __syncthreads();  // the staging area is reused by the next round
    li0 = li1;
  }
  if (WANT_DPOS) {
    float wxg, wyg, wzg;
    tn_contract_bwd(c, dpx, dpy, dpz, wxg, wyg, wzg);
    if (!live) { wxg = wyg = wzg = 0.0f; }
    const float tm = (st + en) / 2.0f;
    float v[6] = {wxg, wyg, wzg, wxg * tm, wyg * tm, wzg * tm};
    // patch order: lanes 4 apart are consecutive depths of one ray when the group holds 4 rays
    const int r32 = (int)ray;
    const int lead = __shfl(r32, lane & 3, 64);
    if (__all(r32 == lead)) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        float r = v[k];
        r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 8, 64);
        r += __shfl_xor(r, 16, 64);
        r += __shfl_xor(r, 32, 64);
        if (lane < 4 && r != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), r);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (v[k] != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), v[k]);
    }
  }
}

#ifndef FOLD_THREADS
#define FOLD_THREADS 1024  // 2 blocks of 16 waves per CU (64 KB of LDS each).  512: 0.7959 -> 0.7914 ms per step with 1024 (three interleaved pairs); 256: 0.822
#endif
// How the fold sums a bucket's records in LDS: in DOUBLE, with the LDS atomic add.  ds_add_f32 is executed lane by lane on gfx950 (~195
// cycles per wave instruction, conflicts or not); ds_add_f64 is not: ~27 cycles per wave instruction on random slots, integer atomics 11-16
// (scripts/microbench/lds_atomic_rate.hip).  Nothing comes back from the add, so a wave fires its records at the image without waiting for any
// of them, and the table gradient receives the double sum rounded once (order-independent up to that rounding).  Rounds 2-3 used slot LOCKS
// (integer atomic OR to claim, plain read-modify-write, atomic AND to release: three dependent LDS round trips per record, float atomics for
// the contended coarse levels): 141 us for the main grid's 25 M records, where this takes 81.  Block-wide passes separated by barriers are no
// way either: a barrier interval costs ~1 us with 32 waves on the CU whatever it contains (profiles/r02_scatter_alternatives.md).
struct FoldRecs {
  uint32_t idp[4];  // slots of records (2u, 2u+1) as two 16-bit halves
  float2 vals[8];
  uint32_t ok;      // bit j: record j exists and is non-zero
  __device__ __forceinline__ uint32_t id(int j) const { return (j & 1) ? (idp[j >> 1] >> 16) : (idp[j >> 1] & 0xffffu); }
};
__device__ __forceinline__ void fold_load(const uint16_t* __restrict__ ip, const float2* __restrict__ vp, uint32_t first, uint32_t end, FoldRecs& r) {
  // 8 records per thread as 4 pairs: every load instruction of a wave covers one contiguous piece
  r.ok = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t rec = first + u * 128;  // even
    uint32_t iw = 0;
    float4 vq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rec < end) {  // the bucket's capacity is even: the pair never leaves the bucket
      iw = *reinterpret_cast<const uint32_t*>(ip + rec);
      vq = *reinterpret_cast<const float4*>(vp + rec);
    }
    r.idp[u] = iw;
    r.vals[2 * u] = make_float2(vq.x, vq.y);
    r.vals[2 * u + 1] = make_float2(vq.z, vq.w);
    if (rec < end && (vq.x != 0.0f || vq.y != 0.0f)) r.ok |= 1u << (2 * u);
    if (rec + 1 < end && (vq.z != 0.0f || vq.w != 0.0f)) r.ok |= 2u << (2 * u);
  }
}
__global__ void __launch_bounds__(FOLD_THREADS) k_grid_fold(GridK g, BinK bk, uint32_t first_block) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];  // two double images of the bucket's slots: [1 << slice_log2] x, then y
  const uint32_t blk = first_block + blockIdx.x;  // a launch may cover the blocks of a level range only
  uint32_t l = 0;
#pragma unroll 1
  for (int i = 1; i < g.L; ++i)
    if (blk >= bk.blk0[i]) l = i;
  const uint32_t chunk = bk.chunk[l], chunks_per_bucket = (bk.cap + chunk - 1) / chunk;
  // chunk-major inside a level: consecutive blocks = the same chunk of consecutive buckets.  Blocks go to the 8 XCDs round-robin by index, and
  // only the first chunk or two of a bucket hold records: bucket-major order put every live block of the sparse levels on XCDs 0 and 4
  // (4 chunks per bucket), which then ran twice as long as the other six.
  const uint32_t ch = (blk - bk.blk0[l]) / bk.nslices, sl = (blk - bk.blk0[l]) % bk.nslices;
  (void)chunks_per_bucket;
  const uint32_t raw_count = bk.count[(size_t)(l * bk.nslices + sl) * bk.cstride];
  const uint32_t count = min(raw_count, bk.cap);
  // table_grad_is_zero holds for this bucket's slots only if nothing has been added to them since the promise was given: a bucket that overflowed
  // sent its surplus records straight into table_grad (bin pass, float atomics) -- those slots must be added to, not stored
  const bool store = g.grad_zero && raw_count <= bk.cap;
  const uint32_t begin = ch * chunk;
  if (begin >= count) return;  // whole block leaves together
  if (bk.trace && threadIdx.x == 0) bk.trace[16 * blk] = wall_clock64();
  const uint32_t end = min(count, begin + chunk);
  const bool split = count > chunk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t slots = 1u << bk.slice_log2;
  const size_t base = (size_t)l * bk.level_stride + (size_t)sl * bk.cap;  // multiple of 8 records
  const uint16_t* ip = bk.idx + base;
  const float2* vp = bk.val + base;
  const uint32_t mine = wave * 512 + 2 * lane;
  double* ax = reinterpret_cast<double*>(s_mem);  // [slots] first component, then [slots] second component
  double* ay = ax + slots;
#define FOLD_FIRE(REC)                                                                      \
  _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
    if (((REC).ok >> j) & 1u) {                                                               \
      const uint32_t id = (REC).id(j);                                                        \
      if ((REC).vals[j].x != 0.0f) unsafeAtomicAdd(&ax[id], (double)(REC).vals[j].x);         \
      if ((REC).vals[j].y != 0.0f) unsafeAtomicAdd(&ay[id], (double)(REC).vals[j].y);         \
    }                                                                                         \
  }
  FoldRecs cur, nxt;
  fold_load(ip, vp, begin + mine, end, cur);  // in flight while the LDS image is cleared
  {
    float4* z = reinterpret_cast<float4*>(s_mem);
    for (uint32_t t = tid; t < slots; t += FOLD_THREADS) z[t] = make_float4(0.f, 0.f, 0.f, 0.f);  // 4 floats = 16 B per slot
  }
  __syncthreads();
  if (bk.trace && threadIdx.x == 0) bk.trace[16 * blk + 1] = wall_clock64();
  for (uint32_t blk = begin + wave * 512; blk < end; blk += FOLD_THREADS * 8) {  // per wave: no barrier inside
    nxt.ok = 0;
    if (blk + FOLD_THREADS * 8 < end) fold_load(ip, vp, blk + FOLD_THREADS * 8 + 2 * lane, end, nxt);
    FOLD_FIRE(cur)
    cur = nxt;
  }
#undef FOLD_FIRE
  __syncthreads();
  if (bk.trace && threadIdx.x == 0) bk.trace[16 * blk + 2] = wall_clock64();
  float2* dst = g.grad + (size_t)l * g.tsize + ((size_t)sl << bk.slice_log2);
  for (uint32_t t0 = 0; t0 < slots; t0 += FOLD_THREADS * 8) {
    float2 v[8], cv[8];
    bool nz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // every load first (one round trip for the 8 slots of a thread), then the stores
      const uint32_t t = t0 + u * FOLD_THREADS + tid;
      nz[u] = false;
      if (t < slots) {
        v[u] = make_float2((float)ax[t], (float)ay[t]);
        nz[u] = v[u].x != 0.0f || v[u].y != 0.0f;  // untouched slots keep an exactly-zero gradient
        if (nz[u] && !split) cv[u] = store ? make_float2(0.f, 0.f) : dst[t];  // (the caller vouches for zeros: nothing to read back)
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t t = t0 + u * FOLD_THREADS + tid;
      if (!nz[u]) continue;
      if (!split) {
        const float2 r = make_float2(cv[u].x + v[u].x, cv[u].y + v[u].y);
        dst[t] = r;
        // GradScaler's found_inf on the FINAL value (x - x is 0 for finite x, NaN otherwise)
        if (g.nonfinite != nullptr && ((r.x - r.x) + (r.y - r.y)) != 0.0f) *g.nonfinite = 1.0f;
      } else {
        if (v[u].x != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(dst + t), v[u].x);
        if (v[u].y != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(dst + t) + 1, v[u].y);
        // (a bucket split over several blocks: this block's share is checked -- a non-finite share makes the sum non-finite; only a sum of
        // finite shares that overflows would go unseen)
        if (g.nonfinite != nullptr && ((v[u].x - v[u].x) + (v[u].y - v[u].y)) != 0.0f) *g.nonfinite = 1.0f;
      }
    }
  }
  if (bk.trace) {
    __syncthreads();
    if (threadIdx.x == 0) bk.trace[16 * blk + 3] = wall_clock64();
  }
}

// ======================================================================================================================================
// Segmented path (TN_SCATTER_MODE=2): block-private record regions, no counting pass, no reservations.
//
// The binned path above evaluates every level's 8 corner slots TWICE per sample (phase A counts the records per bucket so that phase B can
// stage them in bucket order and one returning global atomic per bucket can reserve their run in the bucket's global array) and its copy-out
// looks every record's bucket up again.  The pass is bound by vector-ALU issue (profiles/r05_experiments.md), and keeping phase A while
// evaluating the slots once loses to the reservation latency (k_grid_bin_p, one level per block: measured, removed).  So the reservation
// goes: a bin block OWNS a region of the record arrays per level -- [level][bin block][BIN_THREADS * 8 records], as many as it can ever
// produce -- writes its records there sorted by bucket (rank = the value an LDS atomic returns while the slots are still in registers, bucket
// offsets = one 128-entry prefix per level) as ONE linear, vectorised copy, and leaves a header {offset, count} per bucket.  The fold block of a
// bucket walks the segments the bin blocks left for it.  No global atomics, no counters to zero, no overflow path (a region cannot overflow).
struct SegK {
  uint16_t* idx;   // [L][NB][SEG_REGION] slot inside the bucket, records of a (level, bin block) sorted by bucket, every bucket's run at an even offset
  float2* val;     // same shape
  uint32_t* hdr;   // [L][nslices][NB]: offset | count << 16 of the bucket's segment inside the (level, bin block) region
  uint32_t NB;     // bin blocks per level (blockIdx.x extent of the bin pass)
  int slice_log2, nslices;
  uint32_t merge_mask;
  uint32_t segc;    // segments (bin blocks) per fold block
  uint32_t chunks;  // fold blocks per bucket = ceil(NB / segc)
};
#define SEG_CAP (BIN_THREADS * 8)
// a block's region of a level: its records, every bucket's run at an EVEN offset (one unused slot behind an odd run) -- the fold reads PAIRS
#define SEG_REGION (SEG_CAP + TN_BIN_MAX_SLICES)
#ifndef SEG_FOLD_ABLATE
#define SEG_FOLD_ABLATE 0
#endif

template <bool WANT_DPOS>
__global__ void __launch_bounds__(BIN_THREADS, 4) k_seg_bin(GridK g, const float* __restrict__ origins, const float* __restrict__ directions,
                                                         const float* __restrict__ e_bins, const float* __restrict__ g_enc, int ld, int64_t N, int S,
                                                         float* __restrict__ d_origins, float* __restrict__ d_directions, int level_groups, SegK sk,
                                                         DposArgs cw, int cw_blocks) {
  __shared__ uint32_t s_cnt[TN_BIN_MAX_SLICES];      // records per bucket of the level being ranked (zero between levels)
  __shared__ uint32_t s_off[TN_BIN_MAX_SLICES + 1];  // exclusive prefix; [nslices] = the level's record count
  __shared__ __attribute__((aligned(16))) uint16_t s_i16[SEG_REGION];
  __shared__ __attribute__((aligned(16))) float2 s_val[SEG_REGION];
  // co-work blocks (the last cw_blocks of every grid row: dispatched between the bin blocks of consecutive level groups): the main field's d
  // position pass, a streaming kernel, beside this pass, which is bound by vector-instruction issue -- in ONE launch, because a second active
  // queue costs an iteration more than the pass takes (tn_field.hip).  (Spread evenly through the rows instead: no gain at all, 0.623 vs
  // 0.616-0.622 ms per step without; at the row ends 0.6115 vs 0.6239.)
  const unsigned bx = blockIdx.x;
  if (!WANT_DPOS && bx >= sk.NB) {
    field_dpos_body(cw, blockIdx.y * (unsigned)cw_blocks + (bx - sk.NB), (unsigned)cw_blocks * gridDim.y, reinterpret_cast<float*>(s_val),
                    reinterpret_cast<int*>(s_cnt));
    return;
  }
  const int lane = tn_lane();
  const int tid = threadIdx.x, wv = tid >> 6;
  const int ns = sk.nslices;
  const int64_t P = N * (int64_t)S;
  int64_t i = (int64_t)bx * BIN_THREADS + tid;
  const bool live = i < P;
  if (!live) i = P - 1;
  int64_t ray;
  int s;
  tn_patch_order(i, N, S, ray, s);
  const int64_t p = ray * S + s;
  const float* o = origins + ray * 3;
  const float* d = directions + ray * 3;
  const float* eb = e_bins + ray * (S + 1) + s;
  const float st = eb[0], en = eb[1];
  const Contracted c = tn_contract(o[0], o[1], o[2], d[0], d[1], d[2], st, en);
  const int nlev = (g.L - (int)blockIdx.y + level_groups - 1) / level_groups;  // levels of this block: blockIdx.y + li * level_groups
  constexpr int GVP = WANT_DPOS ? 0 : 5;  // d enc of the first levels, requested up front (the d-position variant has no registers for it)
  float2 gvp[GVP + 1];
#pragma unroll
  for (int li = 0; li < GVP; ++li) {
    const int l = blockIdx.y + li * level_groups;
    gvp[li] = (live && li < nlev) ? *reinterpret_cast<const float2*>(ld > 0 ? g_enc + p * ld + 2 * l : g_enc + (int64_t)l * 2 * P + 2 * p) : make_float2(0.f, 0.f);
  }
  for (int t = tid; t < ns; t += BIN_THREADS) s_cnt[t] = 0;
  __syncthreads();
  const uint32_t smask = (1u << sk.slice_log2) - 1u;
  float dpx = 0.f, dpy = 0.f, dpz = 0.f;
#pragma unroll 1
  for (int li = 0; li < nlev; ++li) {
    const int l = blockIdx.y + li * level_groups;
    const float res = g.res[l];
    const bool merge = (sk.merge_mask >> l) & 1u;
    BinLevel b;
    bin_level(c, res, g.mask, merge, live, lane, b);
    // rank of every record inside its bucket: the value the counting atomic returns (the slots stay in registers until they are staged)
    // (kept in bits 20+ of the slot's own register -- slots have at most 20 bits on this path, a rank at most 12: no registers of its own)
    if (b.emit) {
#pragma unroll
      for (int k = 0; k < 8; ++k) b.idx[k] |= atomicAdd(&s_cnt[b.idx[k] >> sk.slice_log2], 1u) << 20;
    }
    __syncthreads();  // every record of the level is counted
    if (wv == (li & (BIN_THREADS / 64 - 1))) {
      // one wave (a different one per level): exclusive prefix over the buckets, the headers, and the counters back to zero
      uint32_t run = 0;
      uint32_t* hdr = sk.hdr + ((size_t)l * ns) * sk.NB + bx;
      for (int base = 0; base < ns; base += 64) {
        const int sl = base + lane;
        const uint32_t cnt = sl < ns ? s_cnt[sl] : 0u;
        const uint32_t ce = (cnt + 1u) & ~1u;  // runs start at even offsets and hold an even number of records: the fold reads pairs
        uint32_t inc = ce;
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) {
          const uint32_t t = __shfl_up(inc, o2, 64);
          if (lane >= o2) inc += t;
        }
        if (sl < ns) {
          const uint32_t off = run + inc - ce;
          s_off[sl] = off;
          s_cnt[sl] = 0u;
          hdr[(size_t)sl * sk.NB] = off | (cnt << 16);  // (the REAL count: the fold ignores the second half of an odd run's last pair -- writing a
                                                        // pad record here cost the d-position variant two spilled registers, 5 us on the 1 M-sample grid)
        }
        run += __shfl(inc, 63, 64);
      }
      if (lane == 0) s_off[ns] = run;
    }
    // (everybody, beside the prefix: this level's values)
    float2 gv = make_float2(0.f, 0.f);
    if (li < GVP) {
#pragma unroll
      for (int q = 0; q < GVP; ++q)
        if (q == li) gv = gvp[q];
    } else if (live) {
      gv = *reinterpret_cast<const float2*>(ld > 0 ? g_enc + p * ld + 2 * l : g_enc + (int64_t)l * 2 * P + 2 * p);
    }
    if (WANT_DPOS) {
      // d enc / d position from the corner values: s_k = <g, table[corner k]>, then the three one-sided differences of the trilinear form
      const float2* tb = g.table + (size_t)l * g.tsize;
      float sk8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float2 t = tb[b.idx[k] & 0xfffffu];
        sk8[k] = gv.x * t.x + gv.y * t.y;
      }
      float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          ax += b.wy[u] * b.wz[v] * (sk8[1 + 2 * u + 4 * v] - sk8[0 + 2 * u + 4 * v]);
          ay += b.wx[u] * b.wz[v] * (sk8[u + 2 + 4 * v] - sk8[u + 0 + 4 * v]);
          az += b.wx[u] * b.wy[v] * (sk8[u + 2 * v + 4] - sk8[u + 2 * v + 0]);
        }
      dpx += ax * res;
      dpy += ay * res;
      dpz += az * res;
    }
    float vx[8], vy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float w = b.wx[k & 1] * b.wy[(k >> 1) & 1] * b.wz[k >> 2];
      vx[k] = w * gv.x;
      vy[k] = w * gv.y;
    }
    if (merge && b.maxlen > 1) bin_run_sums(vx, vy, lane, b);  // the run's last lane ends up with the run's sums
    __syncthreads();  // the bucket offsets are known
    if (b.emit) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t pos = s_off[(b.idx[k] & 0xfffffu) >> sk.slice_log2] + (b.idx[k] >> 20);
        s_i16[pos] = (uint16_t)(b.idx[k] & smask);
        s_val[pos] = make_float2(vx[k], vy[k]);
      }
    }
    __syncthreads();  // staged
    // linear copy of the staged records into the block's region of the level: 16 bytes per lane and instruction
    const uint32_t total = s_off[ns];
    const size_t reg = ((size_t)l * sk.NB + bx) * SEG_REGION;
    {
      float4* __restrict__ dv = reinterpret_cast<float4*>(sk.val + reg);
      const float4* sv = reinterpret_cast<const float4*>(s_val);
      for (uint32_t j = tid; j < (total + 1) / 2; j += BIN_THREADS) dv[j] = sv[j];
      uint4* __restrict__ di = reinterpret_cast<uint4*>(sk.idx + reg);
      const uint4* si = reinterpret_cast<const uint4*>(s_i16);
      for (uint32_t j = tid; j < (total + 7) / 8; j += BIN_THREADS) di[j] = si[j];
    }
    // (no barrier here: the next level's ranking touches s_cnt only, which the prefix wave left zeroed before the barrier above; its staging
    // waits behind two more barriers, which nobody passes before everybody has finished this copy)
  }
  if (WANT_DPOS) {
    float wxg, wyg, wzg;
    tn_contract_bwd(c, dpx, dpy, dpz, wxg, wyg, wzg);
    if (!live) { wxg = wyg = wzg = 0.0f; }
    const float tm = (st + en) / 2.0f;
    float v[6] = {wxg, wyg, wzg, wxg * tm, wyg * tm, wzg * tm};
    // patch order: lanes 4 apart are consecutive depths of one ray when the group holds 4 rays
    const int r32 = (int)ray;
    const int lead = __shfl(r32, lane & 3, 64);
    if (__all(r32 == lead)) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        float r = v[k];
        r += __shfl_xor(r, 4, 64);
        r += __shfl_xor(r, 8, 64);
        r += __shfl_xor(r, 16, 64);
        r += __shfl_xor(r, 32, 64);
        if (lane < 4 && r != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), r);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (v[k] != 0.0f) atomicAdd((k < 3 ? d_origins : d_directions) + ray * 3 + (k % 3), v[k]);
    }
  }
}

// Fold of the segmented path: block = (level, bucket, chunk of bin blocks).  It reads the headers the bin blocks left for its bucket, then its
// waves walk the segments -- PACK segments per wave and iteration (64 / PACK lanes each: a segment of a fine main-grid level holds ~32 records,
// of a proposal grid ~60, of a merged coarse level a handful; PACK follows the block's average), lane = record, two loads per record, two
// double-precision LDS adds (see k_grid_fold) -- and the bucket's image goes to the table gradient as in k_grid_fold.
__global__ void __launch_bounds__(FOLD_THREADS, 8) k_seg_fold(GridK g, SegK sk, uint32_t level_begin, uint32_t grad_zero_promise, PoseFinishArgs pf,
                                                           uint32_t pf_blocks) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];  // [slots] double x, [slots] double y, then [segc] headers, [segc] running counts, 16 words
  // co-work blocks (the FIRST pf_blocks, a multiple of 8 so that the fold blocks keep their XCDs): the launch that ends the iteration's backward
  // (tn_pose_finish.h) -- short latency chains beside a pass that is bound by its LDS adds, instead of 9 us in line behind it
  if (blockIdx.x < pf_blocks) {
    if ((int)blockIdx.x < pf.total_blocks) pose_finish_body(pf, (int)blockIdx.x);
    return;
  }
  const uint32_t fb = blockIdx.x - pf_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t ns = (uint32_t)sk.nslices, per_level = ns * sk.chunks;
  const uint32_t l = level_begin + fb / per_level, rel = fb % per_level;
  // chunk-major inside a level.  Blocks go to the 8 XCDs round-robin by index; the segments of buckets s and s+1 of one bin block are neighbours in
  // memory (they share cache lines: a segment of a fine main-grid level is 64 B of slots and 256 B of values at an arbitrary offset), so groups
  // of 4 consecutive buckets are given to ONE XCD, as consecutive blocks of it: the shared lines are then hits in that XCD's L2.
  const uint32_t ch = rel / ns, p = rel % ns;
  const uint32_t sl = (ns % 32u == 0u) ? ((p >> 5) << 5) + ((p & 7u) << 2) + ((p >> 3) & 3u) : p;
  const uint32_t slots = 1u << sk.slice_log2;
  double* ax = reinterpret_cast<double*>(s_mem);
  double* ay = ax + slots;
  uint32_t* s_h = reinterpret_cast<uint32_t*>(s_mem + 4 * slots);
  uint32_t* s_pre = s_h + sk.segc;    // [segc] records before the segment, counted from the first segment of its wave's 64
  uint32_t* s_tot = s_pre + sk.segc;  // [FOLD_THREADS / 64] record counts of the waves' headers
  const uint32_t b0 = ch * sk.segc, nseg = min(sk.segc, sk.NB - b0);
  // the headers (at most FOLD_THREADS: seg_plan) are in flight while the image is cleared
  const uint32_t my_h = (uint32_t)tid < nseg ? sk.hdr[((size_t)l * ns + sl) * sk.NB + b0 + tid] : 0u;
  {
    float4* z = reinterpret_cast<float4*>(s_mem);
    for (uint32_t t = tid; t < slots; t += FOLD_THREADS) z[t] = make_float4(0.f, 0.f, 0.f, 0.f);  // 4 floats = 16 B per slot
  }
  {
    const uint32_t c = ((my_h >> 16) + 1u) >> 1;  // PAIRS of records (the runs start at even offsets; an odd run's last pair is half empty)
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if ((uint32_t)tid < nseg) { s_h[tid] = my_h; s_pre[tid] = inc - c; }
    if (lane == 63) s_tot[wave] = inc;
  }
  __syncthreads();
  uint32_t total = 0;  // pairs
#pragma unroll
  for (int w = 0; w < FOLD_THREADS / 64; ++w) total += s_tot[w];
  if (total == 0) return;  // whole block leaves together
  const bool split = sk.chunks > 1;
  const bool store = grad_zero_promise && !split;
  // The LDS adds are what this kernel's time is made of, and a ds_add_f64 costs the same with 20 lanes enabled as with 64 (one segment per wave
  // instruction, the first version: 768 half-empty adds per main-grid block instead of the binned fold's 384 full ones, 88 us instead of 70 --
  // unchanged by deeper prefetch, hoisted header loads or bucket-to-XCD grouping).  So the lanes are kept full: the block's records, segment after
  // segment, are ONE stream, cut into equal shares for groups of `gw` lanes; a group walks its share gw PAIRS at a time (one 4-byte slot pair +
  // one 16-byte value pair per lane), stepping over segment boundaries in the middle of a step (per lane: segment, offset inside it; a boundary
  // costs the lane one header read from LDS).  Equal shares in RECORDS: with equal shares in segments the slowest of the 64 groups had ~17 % more
  // than the mean.  Pairs: the walk (~40 vector instructions per step) was a quarter of the kernel with one record per lane and step.
  const uint32_t avg = 2u * total / nseg;
  const int gw_log2 = avg >= 256 ? 6 : (avg >= 96 ? 5 : 4);
  const uint32_t gw = 1u << gw_log2, ngroups = FOLD_THREADS >> gw_log2;
  const uint32_t share = (((total + ngroups - 1) / ngroups) + gw - 1) & ~(gw - 1);
  const uint32_t r_begin = ((uint32_t)tid >> gw_log2) * share, r_end = min(total, r_begin + share);
  uint32_t rr = r_begin + ((uint32_t)tid & (gw - 1));  // the lane's next pair of the stream
  const size_t lvl_base = (size_t)l * sk.NB + b0;
  uint32_t sg = 0, t = 0, cnt = 0, cnt_rec = 0;
  size_t base = 0;
  auto open_seg = [&]() {
    const uint32_t h = s_h[sg];
    cnt_rec = h >> 16;
    cnt = (cnt_rec + 1u) >> 1;
    base = (lvl_base + sg) * SEG_REGION + (h & 0xffffu);
  };
  if (r_begin < total) {
    // the segment that holds pair r_begin: first the wave's 64 headers it lies in, then a binary search on their running counts
    uint32_t before = 0, w = 0;
#pragma unroll
    for (int i = 0; i < FOLD_THREADS / 64; ++i) {
      const uint32_t c = s_tot[i];
      if (i == (int)w && r_begin >= before + c) { before += c; ++w; }
    }
    const uint32_t want = r_begin - before;  // < s_tot[w]
    uint32_t lo = w * 64, hi = min(nseg, lo + 64);  // answer: the last segment in [lo, hi) with s_pre <= want (empty segments in front of it share that value)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s_pre[mid] <= want) lo = mid; else hi = mid;
    }
    sg = lo;
    open_seg();
    t = want - s_pre[sg] + ((uint32_t)tid & (gw - 1));
  }
  struct FoldRec { uint32_t idp; float4 v; bool ok; };
  auto fetch = [&](FoldRec& r) {  // the lane's next pair (ok = false once the share is used up, and from then on)
    r.ok = rr < r_end;
    r.idp = 0;
    r.v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r.ok) {
      while (t >= cnt) {  // (a pair with index < total exists: the walk ends inside the headers)
        t -= cnt;
        ++sg;
        open_seg();
      }
#if SEG_FOLD_ABLATE & 2  // (timing experiments: no record loads)
      r.idp = ((uint32_t)(base + 2 * t) & (slots - 1)) * 0x10001u;
      r.v = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
#else
      r.idp = *reinterpret_cast<const uint32_t*>(sk.idx + base + 2 * (size_t)t);
      r.v = *reinterpret_cast<const float4*>(sk.val + base + 2 * (size_t)t);
#endif
      if (2u * t + 1u >= cnt_rec) { r.idp &= 0xffffu; r.v.z = 0.0f; r.v.w = 0.0f; }  // the unused slot behind an odd run (whatever the staging area held)
    }
    t += gw;
    rr += gw;
  };
#ifndef SEG_FOLD_D
#define SEG_FOLD_D 4
#endif
  constexpr int FOLD_D = SEG_FOLD_D;  // pairs in flight per lane (2 / 6 / 8: see profiles/r05_experiments.md)
  FoldRec q[FOLD_D];
#pragma unroll
  for (int d = 0; d < FOLD_D; ++d) fetch(q[d]);
  while (__any(q[0].ok)) {
#pragma unroll
    for (int d = 0; d < FOLD_D; ++d) {
      const FoldRec c = q[d];
      fetch(q[d]);
#if SEG_FOLD_ABLATE & 1  // (timing experiments: no LDS adds -- the loads stay alive through a store that never happens)
      if (c.ok && c.v.x == 123.456f && c.idp == 77u) ax[c.idp & 0xffffu] = (double)c.v.y;
#else
      // (one test per pair, not one per value: an add of +-0 changes nothing -- untouched slots stay exactly zero -- and an LDS add costs the same
      // with a lane more or less; a step whose pairs are ALL zero is skipped as a whole)
      if (c.ok && ((c.v.x != 0.0f) | (c.v.y != 0.0f) | (c.v.z != 0.0f) | (c.v.w != 0.0f))) {
        const uint32_t i0 = c.idp & 0xffffu, i1 = c.idp >> 16;
        unsafeAtomicAdd(&ax[i0], (double)c.v.x);
        unsafeAtomicAdd(&ay[i0], (double)c.v.y);
        unsafeAtomicAdd(&ax[i1], (double)c.v.z);
        unsafeAtomicAdd(&ay[i1], (double)c.v.w);
      }
#endif
    }
  }
  __syncthreads();
#if SEG_FOLD_ABLATE & 4  // (timing experiments: no flush)
  if (total != 0xffffffffu) return;
#endif
  float2* dst = g.grad + (size_t)l * g.tsize + ((size_t)sl << sk.slice_log2);
  for (uint32_t tt = 0; tt < slots; tt += FOLD_THREADS * 8) {
    float2 w[8], cv[8];
    bool nz[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // every load first (one round trip for the 8 slots of a thread), then the stores
      const uint32_t t = tt + u * FOLD_THREADS + tid;
      nz[u] = false;
      if (t < slots) {
        w[u] = make_float2((float)ax[t], (float)ay[t]);
        nz[u] = w[u].x != 0.0f || w[u].y != 0.0f;  // untouched slots keep an exactly-zero gradient
        if (nz[u] && !split) cv[u] = store ? make_float2(0.f, 0.f) : dst[t];  // (the caller vouches for zeros: nothing to read back)
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t t = tt + u * FOLD_THREADS + tid;
      if (!nz[u]) continue;
      if (!split) {
        const float2 r = make_float2(cv[u].x + w[u].x, cv[u].y + w[u].y);
        dst[t] = r;
        if (g.nonfinite != nullptr && ((r.x - r.x) + (r.y - r.y)) != 0.0f) *g.nonfinite = 1.0f;  // GradScaler's found_inf on the FINAL value
      } else {
        if (w[u].x != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(dst + t), w[u].x);
        if (w[u].y != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(dst + t) + 1, w[u].y);
        if (g.nonfinite != nullptr && ((w[u].x - w[u].x) + (w[u].y - w[u].y)) != 0.0f) *g.nonfinite = 1.0f;
      }
    }
  }
}

static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* e = getenv(name);
  int v = e ? atoi(e) : dflt;
  return v < lo ? lo : (v > hi ? hi : v);
}
// tuning knob (0 disables the replicas)
static int dense_replicas() { static int r = env_int("TN_SCATTER_REPLICAS", 16, 0, 64); return r; }

// Replica plan: dense replicas for the coarse levels, in level order while they qualify and the scratch lasts.  (Whole-level replicas in
// the table's own hashed layout were tried for the small proposal tables: 7 % on that kernel alone, nothing -- slightly negative -- on the
// step, because of the extra zero-fill and fold; removed.)  A replica does not pay when the batch is much smaller than it (the zero-fill and
// the fold are O(replica size)).  Fills everything of rk except rk.rep; returns the scratch entries (float2) the plan uses.  Depends only on
// the grid and P.
static int64_t plan_replicas(const TnGrid& grid, int64_t P, ReplicaK& rk) {
  const int64_t cap = TN_SCATTER_SCRATCH_BYTES / (int64_t)sizeof(float2);
  const int64_t T = 1ll << grid.log2_hashmap_size;
  int64_t used = 0, total = 0;
  if (dense_replicas() <= 0) return 0;
  for (int l = 0; l < grid.num_levels; ++l) {
    float r = grid.res[l];
    int64_t n = 0;
    int kind = TN_REP_NONE, R = 0;
    if (r >= 1.0f && r <= 62.0f) {
      int64_t r1 = (int64_t)ceilf(r) + 1;
      n = r1 * r1 * r1;
      R = dense_replicas();
      if (n <= T && n <= 4 * P && used + n * R <= cap) kind = TN_REP_DENSE;
    }
    if (kind == TN_REP_NONE) break;  // levels are in increasing resolution: no dense level after the first that does not qualify
    rk.kinds |= (uint32_t)kind << (2 * l);
    rk.n[l] = (uint32_t)n;
    rk.off[l] = (uint32_t)used;
    rk.first[l] = (uint32_t)total;
    rk.R[l] = (uint8_t)R;
    used += n * R;
    total += n;
  }
  rk.total = (uint32_t)total;
  return used;
}

// TN_SCATTER_MODE: 2 = segmented (default), 1 = binned (rounds 2-5: per-bucket global arrays with reservations), 0 = atomics with dense replicas
// (the round-1 path; also what the dense data-parallel exchange uses).  Read on every call (tests compare the three in one process); the bin and
// the fold launches of one backward must see the same value.
static int scatter_mode() { return env_int("TN_SCATTER_MODE", 2, 0, 2); }
static bool seg_mode() { return scatter_mode() == 2; }
static int merge_res();

// layout of the segmented path for (grid, P, scratch): a pure function of its arguments (the bin pass and the fold launches of a phased backward
// agree on it without any state).  It lives inside the binned path's scratch: 8 records per sample and level instead of 16.
static int seg_plan(const TnGrid& grid, int64_t P, void* scratch, SegK& sk) {
  const int L = grid.num_levels;
  sk = SegK{};
  sk.slice_log2 = std::min(TN_BIN_SLICE_LOG2, grid.log2_hashmap_size);
  sk.nslices = 1 << (grid.log2_hashmap_size - sk.slice_log2);
  TN_REQUIRE(sk.nslices <= TN_BIN_MAX_SLICES, "tn_grid_scatter: table too large for the segmented path");
  const int64_t NB = tn_cdiv(P, BIN_THREADS);
  sk.NB = (uint32_t)NB;
  const int64_t recs = (int64_t)L * NB * SEG_REGION;
  char* base = reinterpret_cast<char*>(scratch) + 256;
  sk.val = reinterpret_cast<float2*>(base);
  sk.idx = reinterpret_cast<uint16_t*>(base + recs * 8);
  sk.hdr = reinterpret_cast<uint32_t*>(base + recs * 10);
  TN_REQUIRE(256 + recs * 10 + (int64_t)L * sk.nslices * NB * 4 <= tn_scatter_scratch_bytes(P, L), "tn_grid_scatter: segmented layout exceeds the scatter scratch");
  for (int l = 0; l < L; ++l)
    if (grid.res[l] <= (float)merge_res()) sk.merge_mask |= 1u << l;
  // fold blocks: a bucket's bin blocks are cut into chunks of `segc` (<= 1024 = FOLD_THREADS: one header per thread).  A grid with many buckets
  // (the main grid: 16 x 128) gets one block per bucket while the headers fit -- chunks flush with float atomics; a grid with few (the proposal
  // grids: 5 x 32) is cut into chunks of 8 x nslices bin blocks, ~32 K records per fold block (first proposal grid, 2048 bin blocks: 128 per
  // chunk 67.7 us, 256 57.4, 512 66.6; second, 768 bin blocks: 37.3 / 33.1 / 46.5)
  int64_t segc = (int64_t)L * sk.nslices >= 1024 ? 1024 : std::max<int64_t>(64, std::min<int64_t>(1024, 8 * sk.nslices));
  segc = std::min<int64_t>(segc, NB);
  segc = tn_cdiv(NB, tn_cdiv(NB, segc));  // equal chunks
  sk.segc = (uint32_t)segc;
  sk.chunks = (uint32_t)tn_cdiv(NB, segc);
  return TN_OK;
}
static int fold_chunk_sparse() { static int r = env_int("TN_SCATTER_SPARSE_CHUNK", 16384, 1024, 32768) & ~1023; return r; }
// Same-cell runs of consecutive samples are summed before they are written (bin_run_sums) on every level whose resolution is at most this.
// Default: every level.  (Until the end of round 4 the default was 256 -- the finer levels of the main grid were thought to hold too few runs
// to pay for the scan; measured with the scan on DPP: 0.809 -> 0.797-0.803 ms per step early in training, where the samples are still spread
// out, and the runs only get longer as the sampler concentrates them at surfaces.)
static int merge_res() { static int r = env_int("TN_SCATTER_MERGE_RES", 1 << 20, 0, 1 << 20); return r; }

// Layout of the binned scatter for (grid, P, scratch): a pure function of its arguments, so the bin pass and the fold launches of a phased
// backward (tn_grid_scatter_bin / tn_grid_scatter_fold) agree on it without any state.
static int bin_plan(const TnGrid& grid, int64_t P, void* scratch, BinK& bk, uint32_t& nblk) {
  const int L = grid.num_levels;
  bk = BinK{};
  bk.slice_log2 = std::min(TN_BIN_SLICE_LOG2, grid.log2_hashmap_size);
  bk.nslices = 1 << (grid.log2_hashmap_size - bk.slice_log2);
  TN_REQUIRE(bk.nslices <= TN_BIN_MAX_SLICES, "tn_grid_scatter: table too large for the binned path");
  // per level tn_bin_level_records(P) records, split evenly over the buckets
  const int64_t cap = (tn_bin_level_records(P) / bk.nslices) & ~7ll;
  TN_REQUIRE(cap * bk.nslices < (1ll << 32), "tn_grid_scatter: batch too large for the binned path");
  bk.cap = (uint32_t)cap;
  bk.level_stride = (uint32_t)(cap * bk.nslices);
  char* base = reinterpret_cast<char*>(scratch);
  bk.count = reinterpret_cast<uint32_t*>(base);
  bk.cstride = 1;  // one 64-B line per counter (16) was measured 15 % SLOWER on the main grid: more lines to fetch, no gain
  const int64_t cnt_bytes = 256 + (int64_t)L * TN_BIN_MAX_SLICES * 4 * TN_BIN_COUNT_STRIDE;
  bk.val = reinterpret_cast<float2*>(base + cnt_bytes);
  bk.idx = reinterpret_cast<uint16_t*>(base + cnt_bytes + (int64_t)L * tn_bin_level_records(P) * 8);
  nblk = 0;
  for (int l = 0; l < L; ++l) {
    if (grid.res[l] <= (float)merge_res()) bk.merge_mask |= 1u << l;
    // live slots of the level: (res+1)^3 cells hashed into 2^log2T slots; below half of the table the buckets are sparse and hot
    const double r1 = ceil((double)grid.res[l]) + 1.0;
    const double T = (double)(1ll << grid.log2_hashmap_size), live = std::min(r1 * r1 * r1, T);
    // ... or there are very many samples per live slot (the proposal grids: 256 samples per ray): smaller chunks spread such buckets over more CUs
    const bool sparse = r1 * r1 * r1 < 0.5 * T || 8.0 * (double)P > 16.0 * live;
    bk.chunk[l] = sparse ? (uint32_t)fold_chunk_sparse() : 32768u;
    bk.blk0[l] = nblk;
    nblk += (uint32_t)(bk.nslices * tn_cdiv(bk.cap, bk.chunk[l]));
  }
  bk.blk0[L] = nblk;
  return TN_OK;
}

void tn_grid_scatter_counters(const TnGrid& grid, int64_t P, void* scratch, uint32_t** ptr, int* words) {
  *ptr = nullptr;
  *words = 0;
  if (P <= 0 || !tn_grid_scatter_is_binned(grid, P, scratch)) return;
  if (seg_mode()) return;  // the segmented path has no counters (no global atomics at all)
  BinK bk;
  uint32_t nblk;
  if (bin_plan(grid, P, scratch, bk, nblk) != TN_OK) return;
  *ptr = bk.count;
  *words = grid.num_levels * bk.nslices * (int)bk.cstride;
}

int tn_grid_scatter_bin(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld, int64_t N,
                        int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, bool counters_zeroed, const DposArgs* cowork) {
  const int64_t P = N * (int64_t)S;
  const int L = grid.num_levels;
  if (seg_mode()) {
    SegK sk;
    int rc = seg_plan(grid, P, scratch, sk);
    if (rc) return rc;
    GridK gk = make_gridk(grid);
    const int blocks = (int)sk.NB;
    int level_groups = 1;  // enough resident work for every CU: split the levels over blockIdx.y while the batch alone gives fewer than ~6 blocks per CU
    while (level_groups < L && (int64_t)blocks * level_groups < 256 * 6) level_groups *= 2;
    level_groups = std::min(level_groups, L);
    TN_REQUIRE(cowork == nullptr || d_origins == nullptr, "tn_grid_scatter: co-work only beside a bin pass without a d position path of its own");
    if (d_origins != nullptr) {
      hipLaunchKernelGGL(k_seg_bin<true>, dim3(blocks, level_groups), dim3(BIN_THREADS), 0, stream, gk, origins, directions, e_bins, g_enc, ld, N, S, d_origins,
                         d_directions, level_groups, sk, DposArgs{}, 0);
    } else {
      // co-work: one tile of 32 samples per wave, BIN_THREADS / 64 tiles per block, spread over the rows of the grid
      int cw_blocks = 0;
      if (cowork != nullptr) {
        const int64_t tiles = tn_cdiv(cowork->N * (int64_t)cowork->S, 32);
        cw_blocks = (int)std::min<int64_t>(tn_cdiv(tn_cdiv(tiles, BIN_THREADS / 64), level_groups), 2048);
      }
      hipLaunchKernelGGL(k_seg_bin<false>, dim3(blocks + cw_blocks, level_groups), dim3(BIN_THREADS), 0, stream, gk, origins, directions, e_bins, g_enc, ld, N, S,
                         d_origins, d_directions, level_groups, sk, cowork ? *cowork : DposArgs{}, cw_blocks);
    }
    TN_CHECK_LAUNCH("tn_grid_scatter(bin, segmented)");
    return TN_OK;
  }
  TN_REQUIRE(cowork == nullptr, "tn_grid_scatter: co-work needs the segmented path (tn_grid_scatter_takes_cowork)");
  BinK bk;
  uint32_t nblk;
  int rc = bin_plan(grid, P, scratch, bk, nblk);
  if (rc) return rc;
  if (!counters_zeroed) {  // (the field / proposal backward kernels zero them on their way: tn_grid_scatter_counters)
    hipError_t e = hipMemsetAsync(bk.count, 0, (size_t)L * bk.nslices * 4 * bk.cstride, stream);
    TN_REQUIRE(e == hipSuccess, "tn_grid_scatter: memset failed: %s", hipGetErrorString(e));
  }
  GridK gk = make_gridk(grid);
  const int blocks = (int)tn_cdiv(P, BIN_THREADS);
  // enough resident work for every CU: split the levels over blockIdx.y while the batch alone gives fewer than ~6 blocks per CU
  int level_groups = 1;
  while (level_groups < L && ((int64_t)blocks * level_groups < 256 * 6 || tn_cdiv(L, level_groups) * bk.nslices > BIN_MAX_COUNTERS)) level_groups *= 2;
  level_groups = std::min(level_groups, L);
  {  // TN_BIN_LEVEL_GROUPS (diagnostic): force the number of level groups (L = one level per block: the level's corner slots are evaluated once)
    static const int forced = env_int("TN_BIN_LEVEL_GROUPS", 0, 0, TN_MAX_LEVELS);
    if (forced > 0 && tn_cdiv(L, std::min(forced, L)) * bk.nslices <= BIN_MAX_COUNTERS) level_groups = std::min(forced, L);
  }
  if (d_origins != nullptr)
    hipLaunchKernelGGL(k_grid_bin<true>, dim3(blocks, level_groups), dim3(BIN_THREADS), 0, stream, gk, origins, directions, e_bins, g_enc, ld, N, S,
                       d_origins, d_directions, level_groups, bk);
  else
    hipLaunchKernelGGL(k_grid_bin<false>, dim3(blocks, level_groups), dim3(BIN_THREADS), 0, stream, gk, origins, directions, e_bins, g_enc, ld, N, S,
                       d_origins, d_directions, level_groups, bk);
  TN_CHECK_LAUNCH("tn_grid_scatter(bin)");
  return TN_OK;
}

// fold of levels [level_begin, level_end) of a grid whose records tn_grid_scatter_bin has written (same grid, P and scratch)
int tn_grid_scatter_fold(const TnGrid& grid, int64_t P, void* scratch, int level_begin, int level_end, hipStream_t stream, const PoseFinishArgs* cowork) {
  const int L = grid.num_levels;
  TN_REQUIRE(level_begin >= 0 && level_begin < level_end && level_end <= L, "tn_grid_scatter_fold: bad level range");
  TN_REQUIRE(cowork == nullptr || seg_mode(), "tn_grid_scatter_fold: co-work needs the segmented path (tn_grid_scatter_takes_cowork)");
  if (seg_mode()) {
    SegK sk;
    int rc = seg_plan(grid, P, scratch, sk);
    if (rc) return rc;
    GridK gk = make_gridk(grid);
    const size_t shmem = (size_t)(2u << sk.slice_log2) * sizeof(double) + (size_t)sk.segc * 8 + (FOLD_THREADS / 64) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_fold), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)((2u << TN_BIN_SLICE_LOG2) * sizeof(double) + 1024 * 8 + (FOLD_THREADS / 64) * 4));
    const uint32_t nblk = (uint32_t)(level_end - level_begin) * (uint32_t)sk.nslices * sk.chunks;
    const uint32_t pfb = cowork ? ((uint32_t)cowork->total_blocks + 7u) & ~7u : 0u;
    hipLaunchKernelGGL(k_seg_fold, dim3(nblk + pfb), dim3(FOLD_THREADS), shmem, stream, gk, sk, (uint32_t)level_begin, (uint32_t)(gk.grad_zero ? 1 : 0),
                       cowork ? *cowork : PoseFinishArgs{}, pfb);
    TN_CHECK_LAUNCH("tn_grid_scatter(fold, segmented)");
    return TN_OK;
  }
  BinK bk;
  uint32_t nblk_all;
  int rc = bin_plan(grid, P, scratch, bk, nblk_all);
  if (rc) return rc;
  GridK gk = make_gridk(grid);
  const uint32_t first_block = bk.blk0[level_begin], nblk = bk.blk0[level_end] - bk.blk0[level_begin];
  const size_t shmem = (size_t)(2u << bk.slice_log2) * sizeof(double);
  // per launch, not once per process: the attribute is per device (a process that drives a second GPU would otherwise fail the fold there)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grid_fold), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)((2u << TN_BIN_SLICE_LOG2) * sizeof(double)));
  static int trace_on = env_int("TN_FOLD_TRACE", 0, 0, 1);
  static unsigned long long* trace_buf = nullptr;
  if (trace_on) {  // diagnostics only: synchronises and prints
    if (!trace_buf) (void)hipMalloc(&trace_buf, (size_t)1 << 22);
    (void)hipMemsetAsync(trace_buf, 0, (size_t)nblk_all * 128, stream);
    bk.trace = trace_buf;
  }
  hipLaunchKernelGGL(k_grid_fold, dim3(nblk), dim3(FOLD_THREADS), shmem, stream, gk, bk, first_block);
  if (trace_on) {
    std::vector<unsigned long long> h((size_t)nblk_all * 16);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), trace_buf, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (uint32_t b = 0; b < nblk_all; ++b)
      if (h[16 * b]) { t0 = std::min(t0, h[16 * b]); t1 = std::max(t1, h[16 * b + 3]); }
    fprintf(stderr, "[fold trace] %u blocks, span %.1f us (wall clock 100 MHz)\n", nblk, (t1 - t0) / 100.0);
    if (const char* dump = getenv("TN_FOLD_TRACE_FILE")) {
      if (FILE* f = fopen(dump, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    for (int l = 0; l < L; ++l) {
      double a = 0, b2 = 0, c = 0, first = 1e30, last = 0; int n = 0;
      double st[9] = {0};
      for (uint32_t b = bk.blk0[l]; b < bk.blk0[l + 1]; ++b) {
        const unsigned long long* q = &h[16 * b];
        if (!q[0]) continue;
        ++n; a += (q[1] - q[0]) / 100.0; b2 += (q[2] - q[1]) / 100.0; c += (q[3] - q[2]) / 100.0;
        first = std::min(first, (q[0] - t0) / 100.0); last = std::max(last, (q[3] - t0) / 100.0);
        unsigned long long prev = q[1];
        for (int k = 0; k < 9; ++k) { if (q[4 + k]) { st[k] += (q[4 + k] - prev) / 100.0; prev = q[4 + k]; } }
      }
      if (n) fprintf(stderr, "  level %2d: %4d active blocks, mean us: setup %.1f passes %.1f flush %.1f; first start %.1f last end %.1f | 1st iteration: passes %.2f %.2f %.2f %.2f retries %.2f %.2f %.2f sync %.2f float-atomics %.2f\n", l, n, a / n, b2 / n, c / n, first, last,
                     st[0] / n, st[1] / n, st[2] / n, st[3] / n, st[4] / n, st[5] / n, st[6] / n, st[7] / n, st[8] / n);
    }
  }
  TN_CHECK_LAUNCH("tn_grid_scatter(fold)");
  return TN_OK;
}

bool tn_grid_scatter_is_binned(const TnGrid& grid, int64_t P, const void* scratch) {
  return scratch != nullptr && scatter_mode() >= 1 && grid.log2_hashmap_size - TN_BIN_SLICE_LOG2 <= 8 && P * 8 < (1ll << 31);
}

bool tn_grid_scatter_takes_cowork(const TnGrid& grid, int64_t P, void* scratch) { return seg_mode() && tn_grid_scatter_is_binned(grid, P, scratch); }

static int grid_scatter_binned(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld,
                               int64_t N, int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, bool counters_zeroed,
                               const DposArgs* cowork, const PoseFinishArgs* fold_cowork) {
  int rc = tn_grid_scatter_bin(grid, origins, directions, e_bins, g_enc, ld, N, S, d_origins, d_directions, scratch, stream, counters_zeroed, cowork);
  if (rc) return rc;
  return tn_grid_scatter_fold(grid, N * (int64_t)S, scratch, 0, grid.num_levels, stream, fold_cowork);
}

int tn_grid_scatter_launch(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld,
                           int64_t N, int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, float* dense_sum,
                           bool counters_zeroed, const DposArgs* cowork, const PoseFinishArgs* fold_cowork) {
  TN_REQUIRE(grid.table && grid.table_grad && origins && directions && e_bins && g_enc, "tn_grid_scatter: null pointer");
  TN_REQUIRE(grid.num_levels >= 1 && grid.num_levels <= TN_MAX_LEVELS && (ld >= 2 * grid.num_levels || ld == TN_LD_LEVEL_MAJOR),
             "tn_grid_scatter: bad level count / row stride");
  int64_t P = N * (int64_t)S;
  if (P == 0) return TN_OK;
  if (dense_sum == nullptr && tn_grid_scatter_is_binned(grid, P, scratch))
    return grid_scatter_binned(grid, origins, directions, e_bins, g_enc, ld, N, S, d_origins, d_directions, scratch, stream, counters_zeroed, cowork, fold_cowork);
  TN_REQUIRE(cowork == nullptr && fold_cowork == nullptr, "tn_grid_scatter: co-work needs the segmented path (tn_grid_scatter_takes_cowork)");
  int grid_dim = (int)std::min<int64_t>(tn_cdiv(P, 64), 256 * 32);
  // resident capacity is 256 CUs x 8 blocks: when the items do not fill a whole number of rounds, split the levels into 2 interleaved groups
  int level_groups = 1;
  if (grid.num_levels >= 8) {
    double rounds = (double)grid_dim / 2048.0;
    if (rounds < 4.0 && (rounds - floor(rounds)) > 0.0 && (rounds - floor(rounds)) < 0.75) level_groups = 2;
  }
  ReplicaK rk{};
  int64_t used = 0;
  if (scratch != nullptr) {
    used = plan_replicas(grid, P, rk);
    rk.rep = reinterpret_cast<float2*>(scratch);
    if (rk.kinds) {
      hipError_t e = hipMemsetAsync(scratch, 0, (size_t)used * sizeof(float2), stream);
      TN_REQUIRE(e == hipSuccess, "tn_grid_scatter: memset failed: %s", hipGetErrorString(e));
    }
  }
  if (dense_sum != nullptr) {
    bool all_dense = scratch != nullptr;
    for (int l = 0; l < grid.num_levels; ++l) all_dense = all_dense && (((rk.kinds >> (2 * l)) & 3u) == TN_REP_DENSE);
    TN_REQUIRE(all_dense, "tn_grid_scatter: dense_sum wanted but not every level of the range is accumulated densely");
  }
  GridK gk = make_gridk(grid);
  hipLaunchKernelGGL(k_grid_scatter, dim3(grid_dim, level_groups), dim3(256), 0, stream, gk, origins, directions, e_bins, g_enc, ld, N, S, d_origins,
                     d_directions, level_groups, rk);
  TN_CHECK_LAUNCH("tn_grid_scatter");
  if (rk.kinds) {
    hipLaunchKernelGGL(k_replica_reduce, dim3((unsigned)tn_cdiv(rk.total, 256)), dim3(256), 0, stream, gk, rk, reinterpret_cast<float2*>(dense_sum));
    TN_CHECK_LAUNCH("tn_grid_scatter(reduce)");
  }
  return TN_OK;
}

// number of dense cells (float2 entries) of a grid whose levels are ALL accumulated densely for `num_points` samples, else 0
int64_t tn_grid_dense_count(const TnGrid& grid, int64_t P) {
  if (P <= 0 || grid.num_levels < 1 || grid.num_levels > TN_MAX_LEVELS) return 0;
  ReplicaK rk{};
  (void)plan_replicas(grid, P, rk);
  for (int l = 0; l < grid.num_levels; ++l)
    if (((rk.kinds >> (2 * l)) & 3u) != TN_REP_DENSE) return 0;
  return (int64_t)rk.total;
}

// dense per-cell sums (as k_replica_reduce hands them over) -> hashed table gradient; thread = one cell
__global__ void __launch_bounds__(256) k_dense_fold(GridK g, ReplicaK rk, const float2* __restrict__ dense_sum) {
  uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rk.total) return;
  float2 v = dense_sum[e];
  if (v.x == 0.0f && v.y == 0.0f) return;
  int l = 0;
#pragma unroll 1
  for (int i = 0; i < g.L; ++i)
    if (rk.first[i] <= e) l = i;
  const uint32_t t = e - rk.first[l];
  const uint32_t r1 = (uint32_t)(int)ceilf(g.res[l]) + 1u;
  uint32_t x = t % r1, y = (t / r1) % r1, z = t / (r1 * r1);
  uint32_t idx = ((x ^ (y * TN_PRIME_Y) ^ (z * TN_PRIME_Z)) & g.mask) + (uint32_t)l * g.tsize;
  float* dst = reinterpret_cast<float*>(g.grad + idx);
  if (v.x != 0.0f) unsafeAtomicAdd(dst, v.x);
  if (v.y != 0.0f) unsafeAtomicAdd(dst + 1, v.y);
  if (g.nonfinite != nullptr && ((v.x - v.x) + (v.y - v.y)) != 0.0f) *g.nonfinite = 1.0f;
}
int tn_grid_dense_fold(const TnGrid& grid, int64_t P, const float* dense_sum, hipStream_t stream) {
  TN_REQUIRE(grid.table_grad && dense_sum, "tn_grid_dense_fold: null pointer");
  ReplicaK rk{};
  (void)plan_replicas(grid, P, rk);
  for (int l = 0; l < grid.num_levels; ++l)
    TN_REQUIRE(((rk.kinds >> (2 * l)) & 3u) == TN_REP_DENSE, "tn_grid_dense_fold: level %d of the range is not accumulated densely", l);
  hipLaunchKernelGGL(k_dense_fold, dim3((unsigned)tn_cdiv(rk.total, 256)), dim3(256), 0, stream, make_gridk(grid), rk, reinterpret_cast<const float2*>(dense_sum));
  TN_CHECK_LAUNCH("tn_grid_dense_fold");
  return TN_OK;
}

extern "C" int64_t tn_hash_scatter_workspace_bytes(int64_t num_points, int32_t num_levels) {
  if (num_points < 0 || num_levels < 1 || num_levels > TN_MAX_LEVELS) return TN_EINVAL;
  return tn_scatter_scratch_bytes(num_points, num_levels);
}

extern "C" int tn_hash_scatter(const TnGrid* grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int32_t ld,
                               int64_t N, int32_t S, float* d_origins, float* d_directions, void* workspace, int64_t workspace_bytes, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(grid != nullptr, "tn_hash_scatter: null grid");
  TN_REQUIRE(((uintptr_t)workspace % 256) == 0, "tn_hash_scatter: workspace must be 256-byte aligned");
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_hash_scatter: d_origins and d_directions must both be given or both NULL");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_hash_scatter: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(grid->log2_hashmap_size >= 1 && grid->log2_hashmap_size <= 24, "tn_hash_scatter: bad log2_hashmap_size");
  TN_REQUIRE(grid->num_levels >= 1 && grid->num_levels <= TN_MAX_LEVELS, "tn_hash_scatter: bad level count");
  TN_REQUIRE(workspace == nullptr || workspace_bytes >= tn_scatter_scratch_bytes(N * (int64_t)S, grid->num_levels),
             "tn_hash_scatter: workspace of %lld bytes, tn_hash_scatter_workspace_bytes(%lld, %d) = %lld", (long long)workspace_bytes, (long long)(N * (int64_t)S),
             grid->num_levels, (long long)tn_scatter_scratch_bytes(N * (int64_t)S, grid->num_levels));
  return tn_grid_scatter_launch(*grid, origins, directions, e_bins, g_enc, ld, N, S, d_origins, d_directions, workspace, tn_s(stream), nullptr);
}
