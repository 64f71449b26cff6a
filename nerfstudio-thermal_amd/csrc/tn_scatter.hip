// Trilinear scatter-add of d(encoding) into a hash table's gradient (+ d position), shared by the proposal grids and the main grid.
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
#include <stdlib.h>

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
      gvs[li] = (live && l < g.L) ? g_enc[p * ld + 2 * l + ft] : 0.0f;
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
        if (tail && v != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(base + idx) + ft, v);
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

int tn_grid_scatter_launch(const TnGrid& grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int ld,
                           int64_t N, int S, float* d_origins, float* d_directions, void* scratch, hipStream_t stream, float* dense_sum) {
  TN_REQUIRE(grid.table && grid.table_grad && origins && directions && e_bins && g_enc, "tn_grid_scatter: null pointer");
  TN_REQUIRE(grid.num_levels >= 1 && grid.num_levels <= TN_MAX_LEVELS && ld >= 2 * grid.num_levels, "tn_grid_scatter: bad level count / row stride");
  int64_t P = N * (int64_t)S;
  if (P == 0) return TN_OK;
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

extern "C" int64_t tn_hash_scatter_workspace_bytes(void) { return TN_SCATTER_SCRATCH_BYTES; }

extern "C" int tn_hash_scatter(const TnGrid* grid, const float* origins, const float* directions, const float* e_bins, const float* g_enc, int32_t ld,
                               int64_t N, int32_t S, float* d_origins, float* d_directions, void* workspace, tn_stream_t stream) {
  if (N == 0) return TN_OK;  // empty batches are valid and touch nothing
  TN_REQUIRE(grid != nullptr, "tn_hash_scatter: null grid");
  TN_REQUIRE(((uintptr_t)workspace % 256) == 0, "tn_hash_scatter: workspace must be 256-byte aligned");
  TN_REQUIRE((d_origins == nullptr) == (d_directions == nullptr), "tn_hash_scatter: d_origins and d_directions must both be given or both NULL");
  TN_REQUIRE(N >= 0 && S >= 1 && S <= TN_MAX_SAMPLES, "tn_hash_scatter: bad N=%lld S=%d", (long long)N, S);
  TN_REQUIRE(grid->log2_hashmap_size >= 1 && grid->log2_hashmap_size <= 24, "tn_hash_scatter: bad log2_hashmap_size");
  return tn_grid_scatter_launch(*grid, origins, directions, e_bins, g_enc, ld, N, S, d_origins, d_directions, workspace, tn_s(stream), nullptr);
}
