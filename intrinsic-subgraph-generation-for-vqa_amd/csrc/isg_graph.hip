// Graph plan kernels: per-graph node ranges and the CSR-by-destination of a PyG COO edge_index.
#include "isg_common.hpp"

#include <algorithm>
#include <string>

namespace isg {

static thread_local std::string g_last_error;

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ISG_OK;
  g_last_error = hipGetErrorString(e);
  return ISG_ELAUNCH;
}

// ---- ptr / nmax -----------------------------------------------------------------------------------
// batch is sorted ascending.  Node n owns the boundaries of every graph id in (batch[n-1], batch[n]];
// the last node also closes all graphs above batch[N-1] (empty trailing graphs).
__global__ void graph_ptr_kernel(const int64_t *__restrict__ batch, int N, int B, int *__restrict__ ptr) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int b = (int)batch[n];
  int prev = n > 0 ? (int)batch[n - 1] : -1;
  if (b >= B) b = B - 1;  // malformed input is clamped, never written out of bounds
  for (int g = prev + 1; g <= b; ++g) ptr[g] = n;
  if (n == N - 1)
    for (int g = b + 1; g <= B; ++g) ptr[g] = N;
}

__global__ void graph_nmax_kernel(const int *__restrict__ ptr, int B, int *__restrict__ nmax) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  int v = 0;
  if (g < B) v = ptr[g + 1] - ptr[g];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(nmax, v);
}

__global__ void fill_i32_kernel(int *__restrict__ p, int n, int v) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void graph_eptr_kernel(const int *__restrict__ ptr, const int *__restrict__ rowptr, int B,
                                  int *__restrict__ eptr, int *__restrict__ emax) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  int cnt = 0;
  if (g <= B) {
    const int lo = rowptr[ptr[g]];
    eptr[g] = lo;
    if (g < B) cnt = rowptr[ptr[g + 1]] - lo;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt = max(cnt, __shfl_xor(cnt, off, 64));
  if (emax && (threadIdx.x & 63) == 0 && cnt > 0) atomicMax(emax, cnt);
}

// ---- CSR build ------------------------------------------------------------------------------------
__global__ void csr_hist_kernel(const int64_t *__restrict__ dst, int E, int N, int *__restrict__ deg) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int d = (int)dst[e];
  if (d >= 0 && d < N) atomicAdd(&deg[d], 1);
}

// Exclusive scan of deg[0..N) into rowptr[0..N] in three small launches: per-chunk sums (1024 elements per
// workgroup), a one-workgroup scan of the chunk sums, per-chunk exclusive scan + offset.
constexpr int SCAN_CHUNK = 1024;

__device__ __forceinline__ int block_inclusive_scan_256(int v, int *s_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  if (lane == 63) s_wave[wave] = v;
  __syncthreads();
  int add = 0;
  for (int w = 0; w < wave; ++w) add += s_wave[w];
  __syncthreads();
  return v + add;
}

__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int *__restrict__ deg, int N, int *__restrict__ sums) {
  __shared__ int s_wave[4];
  const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
  int v = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (base + u < N) v += deg[base + u];
  const int inc = block_inclusive_scan_256(v, s_wave);
  if (threadIdx.x == 255) sums[blockIdx.x] = inc;
}

// one workgroup: exclusive scan of up to 256*per chunk sums, in place; total -> *total_out
__global__ __launch_bounds__(256) void scan_sums_kernel(int *__restrict__ sums, int nchunks, int *__restrict__ total_out) {
  __shared__ int s_wave[4];
  const int per = (nchunks + 255) / 256;
  const int beg = min((int)threadIdx.x * per, nchunks), end = min(beg + per, nchunks);
  int v = 0;
  for (int i = beg; i < end; ++i) v += sums[i];
  const int inc = block_inclusive_scan_256(v, s_wave);
  int run = inc - v;
  for (int i = beg; i < end; ++i) {
    const int t = sums[i];
    sums[i] = run;
    run += t;
  }
  if (threadIdx.x == 255) *total_out = inc;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int *__restrict__ deg, const int *__restrict__ sums, int N,
                                                         int *__restrict__ rowptr) {
  __shared__ int s_wave[4];
  const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
  int d[4];
  int v = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    d[u] = base + u < N ? deg[base + u] : 0;
    v += d[u];
  }
  const int inc = block_inclusive_scan_256(v, s_wave);
  int run = sums[blockIdx.x] + inc - v;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (base + u < N) rowptr[base + u] = run;
    run += d[u];
  }
}

__global__ void csr_fill_kernel(const int64_t *__restrict__ dst, int E, int N, const int *__restrict__ rowptr,
                                int *__restrict__ cursor, int *__restrict__ eid_tmp) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int d = (int)dst[e];
  if (d < 0 || d >= N) return;
  int slot = atomicAdd(&cursor[d], 1);
  eid_tmp[rowptr[d] + slot] = e;
}

// The atomic fill leaves each segment in arrival order; rank every slot by its edge id inside its
// segment so the final order is ascending edge id (= torch_scatter's CPU accumulation order).
__global__ void csr_rank_kernel(const int64_t *__restrict__ edge_index, int E, int N,
                                const int *__restrict__ rowptr, const int *__restrict__ eid_tmp,
                                int *__restrict__ eid, int *__restrict__ src, int *__restrict__ dst,
                                const int *__restrict__ bounds = nullptr, int *bounds_host = nullptr) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (bounds_host && t < 2) {           // the plan's last launch: the bounds are final (isg_graph_plan_build)
    bounds_host[t] = bounds[t];
    __threadfence_system();
  }
  int total = rowptr[N];
  if (t >= total) return;
  int e = eid_tmp[t];
  int d = (int)edge_index[(int64_t)E + e];
  int rb = rowptr[d], re = rowptr[d + 1];
  int rank = 0;
  for (int u = rb; u < re; ++u) rank += (eid_tmp[u] < e) ? 1 : 0;
  int s = (int)edge_index[e];
  s = min(max(s, 0), N - 1);  // a malformed source id must never turn into an out-of-bounds row read
  eid[rb + rank] = e;
  src[rb + rank] = s;
  if (dst) dst[rb + rank] = d;
}

// ---- the whole plan in seven launches (isg_graph_plan_build) --------------------------------------------------------------
// The step-by-step entry points above cost 14 launches of 4-5 us each per batch (6 % of a BASELINE configs[1] step); here the
// independent pieces share a launch: (1) zero the degree / cursor arrays and the bounds + graph boundaries from `batch`,
// (2) in-degree histogram + largest graph, (3) per-chunk degree sums, (4) scan: every workgroup adds up the chunk sums before
// its own (at most a few hundred values) instead of a separate one-workgroup scan, (5) atomic fill + per-graph slot ranges and
// the largest edge count, (6) rank by edge id inside each segment.  (7) is the tile plan (isg_tile_plan).
__global__ void copy_bounds_kernel(const int *__restrict__ bounds, int *bounds_host) {
  if (threadIdx.x < 2) bounds_host[threadIdx.x] = bounds[threadIdx.x];
  __threadfence_system();
}
__global__ void plan_init_kernel(const int64_t *__restrict__ batch, int N, int B, int *__restrict__ ptr, int *__restrict__ zero,
                                 int nzero, int *__restrict__ bounds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nzero) zero[i] = 0;
  if (i < 2) bounds[i] = 0;
  if (N == 0) {
    if (i <= B) ptr[i] = 0;
    return;
  }
  if (i < N) {
    int b = (int)batch[i];
    const int prev = i > 0 ? (int)batch[i - 1] : -1;
    if (b >= B) b = B - 1;
    for (int g = prev + 1; g <= b; ++g) ptr[g] = i;
    if (i == N - 1)
      for (int g = b + 1; g <= B; ++g) ptr[g] = N;
  }
}

__global__ void plan_hist_kernel(const int64_t *__restrict__ dst, int E, int N, int *__restrict__ deg, const int *__restrict__ ptr,
                                 int B, int *__restrict__ nmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) {
    const int d = (int)dst[i];
    if (d >= 0 && d < N) atomicAdd(&deg[d], 1);
  }
  int v = 0;
  if (i < B) v = ptr[i + 1] - ptr[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(nmax, v);
}

__global__ __launch_bounds__(256) void scan_apply_self_kernel(const int *__restrict__ deg, const int *__restrict__ sums, int N,
                                                              int nchunks, int *__restrict__ rowptr) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  int part = 0;      // sum of the chunk sums before this workgroup's chunk (and, in the last workgroup, of all of them)
  for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) part += sums[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) s_base = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  const int base0 = s_base;
  __syncthreads();
  const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
  int d[4];
  int v = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    d[u] = base + u < N ? deg[base + u] : 0;
    v += d[u];
  }
  const int inc = block_inclusive_scan_256(v, s_wave);
  int run = base0 + inc - v;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (base + u < N) rowptr[base + u] = run;
    run += d[u];
  }
  if ((int)blockIdx.x == nchunks - 1 && threadIdx.x == 255) rowptr[N] = base0 + inc;      // number of valid edges
}

__global__ void plan_fill_kernel(const int64_t *__restrict__ dst, int E, int N, const int *__restrict__ rowptr,
                                 int *__restrict__ cursor, int *__restrict__ eid_tmp, const int *__restrict__ ptr, int B,
                                 int *__restrict__ eptr, int *__restrict__ emax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < E) {
    const int d = (int)dst[i];
    if (d >= 0 && d < N) {
      const int slot = atomicAdd(&cursor[d], 1);
      eid_tmp[rowptr[d] + slot] = i;
    }
  }
  int cnt = 0;
  if (i <= B) {
    const int lo = rowptr[ptr[i]];
    eptr[i] = lo;
    if (i < B) cnt = rowptr[ptr[i + 1]] - lo;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt = max(cnt, __shfl_xor(cnt, off, 64));
  if ((threadIdx.x & 63) == 0 && cnt > 0) atomicMax(emax, cnt);
}


// ---- the whole plan as ONE workgroup (small batches) ----------------------------------------------------------------------------
// The six launches above are ~4.5 us each whatever the batch; for a handful of graphs (run_token_coo.py:49-79 evaluates one question
// per forward) they are 27 us of a forward whose GPU time is 0.75 ms, and six of the ~60 launches the host issues on the graph side.
// Below 2 048 nodes / 8 192 edges one workgroup of 1 024 threads walks the same six phases with a barrier between them (every
// array stays where the six-launch form has it; a phase's writes to global memory are visible to the workgroup's other threads
// behind __syncthreads()): identical outputs (the final order inside a segment is by edge id either way).
constexpr int PS_THREADS = 1024, PS_MAX_N = 2048, PS_MAX_E = 8192;
__global__ __launch_bounds__(PS_THREADS) void plan_small_kernel(const int64_t *__restrict__ batch, const int64_t *__restrict__ edge_index,
                                                                int N, int E, int B, int *__restrict__ ptr, int *__restrict__ bounds,
                                                                int *bounds_host, int *__restrict__ rowptr, int *__restrict__ eid,
                                                                int *__restrict__ src, int *__restrict__ dst, int *__restrict__ eptr,
                                                                int *__restrict__ deg, int *__restrict__ cursor, int *__restrict__ eid_tmp) {
  __shared__ int s_scan[PS_THREADS];
  __shared__ int s_max[2];
  const int tid = threadIdx.x;
  // (1) zero; graph boundaries from the sorted batch vector
  for (int i = tid; i <= N; i += PS_THREADS) { deg[i] = 0; cursor[i] = 0; }
  if (tid < 2) s_max[tid] = 0;
  if (N == 0) {
    for (int i = tid; i <= B; i += PS_THREADS) ptr[i] = 0;
  } else {
    for (int i = tid; i < N; i += PS_THREADS) {
      int b = (int)batch[i];
      const int prev = i > 0 ? (int)batch[i - 1] : -1;
      if (b >= B) b = B - 1;
      for (int g = prev + 1; g <= b; ++g) ptr[g] = i;
      if (i == N - 1)
        for (int g = b + 1; g <= B; ++g) ptr[g] = N;
    }
  }
  __syncthreads();
  // (2) in-degree histogram; the largest graph
  const int64_t *dsts = edge_index + E;
  for (int i = tid; i < E; i += PS_THREADS) {
    const int d = (int)dsts[i];
    if (d >= 0 && d < N) atomicAdd(&deg[d], 1);
  }
  {
    int v = 0;
    for (int g = tid; g < B; g += PS_THREADS) v = max(v, ptr[g + 1] - ptr[g]);
    if (v > 0) atomicMax(&s_max[0], v);
  }
  __syncthreads();
  // (3) + (4) exclusive scan of the degrees -> rowptr: a run of ceil(N / threads) nodes per thread, the runs' sums scanned in LDS
  const int per = (N + PS_THREADS - 1) / PS_THREADS;
  const int lo = min(tid * per, N), hi = min(lo + per, N);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += deg[i];
  s_scan[tid] = sum;
  __syncthreads();
  for (int off = 1; off < PS_THREADS; off <<= 1) {
    const int v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  {
    int run = s_scan[tid] - sum;
    for (int i = lo; i < hi; ++i) {
      rowptr[i] = run;
      run += deg[i];
    }
    if (tid == PS_THREADS - 1) rowptr[N] = s_scan[tid];      // number of valid edges
  }
  __syncthreads();
  // (5) atomic fill; per-graph slot ranges and the largest edge count
  for (int i = tid; i < E; i += PS_THREADS) {
    const int d = (int)dsts[i];
    if (d >= 0 && d < N) {
      const int slot = atomicAdd(&cursor[d], 1);
      eid_tmp[rowptr[d] + slot] = i;
    }
  }
  {
    int cnt = 0;
    for (int g = tid; g <= B; g += PS_THREADS) {
      const int l = rowptr[ptr[g]];
      eptr[g] = l;
      if (g < B) cnt = max(cnt, rowptr[ptr[g + 1]] - l);
    }
    if (cnt > 0) atomicMax(&s_max[1], cnt);
  }
  __syncthreads();
  // (6) rank by edge id inside each segment; the bounds
  if (tid < 2) {
    bounds[tid] = s_max[tid];
    if (bounds_host) {
      bounds_host[tid] = s_max[tid];
      __threadfence_system();
    }
  }
  const int total = rowptr[N];
  for (int t = tid; t < total; t += PS_THREADS) {
    const int e = eid_tmp[t];
    const int d = (int)dsts[e];
    const int rb = rowptr[d], re = rowptr[d + 1];
    int rank = 0;
    for (int u = rb; u < re; ++u) rank += (eid_tmp[u] < e) ? 1 : 0;
    int sn = (int)edge_index[e];
    sn = min(max(sn, 0), N - 1);
    eid[rb + rank] = e;
    src[rb + rank] = sn;
    if (dst) dst[rb + rank] = d;
  }
}

}  // namespace isg

using namespace isg;

extern "C" int isg_abi_version(void) { return ISG_ABI_VERSION; }

extern "C" const char *isg_status_string(int status) {
  switch (status) {
    case ISG_OK: return "ok";
    case ISG_EINVAL: return "invalid argument";
    case ISG_EUNSUPPORTED: return "unsupported shape";
    case ISG_ELAUNCH: return "HIP launch failed";
    case ISG_EWORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

extern "C" const char *isg_last_hip_error(void) { return g_last_error.c_str(); }

extern "C" int isg_graph_ptr(const int64_t *batch, int64_t N, int64_t B, int32_t *ptr, int32_t *nmax,
                             void *stream) {
  if (!ptr || !nmax || N < 0 || B < 0 || (N > 0 && !batch)) return ISG_EINVAL;
  if (N >= (1ll << 31) || B >= (1ll << 31)) return ISG_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  // graphs with no node at all (N == 0, or ids never written) must still read as empty ranges
  fill_i32_kernel<<<(int)((B + 1 + 255) / 256), 256, 0, st>>>(ptr, (int)B + 1, 0);
  fill_i32_kernel<<<1, 64, 0, st>>>(nmax, 1, 0);
  if (N > 0 && B > 0) {
    graph_ptr_kernel<<<(int)((N + 255) / 256), 256, 0, st>>>(batch, (int)N, (int)B, ptr);
    graph_nmax_kernel<<<(int)((B + 255) / 256), 256, 0, st>>>(ptr, (int)B, nmax);
  }
  return check_launch();
}

extern "C" size_t isg_csr_workspace_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0) return 0;
  return (size_t)(2 * (N + 1) + E + (N + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * sizeof(int32_t);
}

extern "C" int isg_csr_build(const int64_t *edge_index, int64_t N, int64_t E, int32_t *rowptr, int32_t *eid,
                             int32_t *src, int32_t *dst, void *workspace, size_t workspace_bytes, void *stream) {
  if (!rowptr || N < 0 || E < 0 || (E > 0 && (!edge_index || !eid || !src))) return ISG_EINVAL;
  if (N >= (1ll << 31) || E >= (1ll << 31)) return ISG_EUNSUPPORTED;
  if (!workspace || workspace_bytes < isg_csr_workspace_bytes(N, E)) return ISG_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  int *deg = (int *)workspace;            // N+1
  int *cursor = deg + (N + 1);            // N+1
  int *eid_tmp = cursor + (N + 1);        // E
  int *chunk_sums = eid_tmp + E;          // ceil(N / SCAN_CHUNK) + 1
  const int n = (int)N, e = (int)E;
  fill_i32_kernel<<<(2 * (n + 1) + 255) / 256, 256, 0, st>>>(deg, 2 * (n + 1), 0);
  if (e > 0) csr_hist_kernel<<<(e + 255) / 256, 256, 0, st>>>(edge_index + E, e, n, deg);
  const int nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if (nchunks > 256 * 4096) return ISG_EUNSUPPORTED;
  if (nchunks > 0) scan_chunk_sums_kernel<<<nchunks, 256, 0, st>>>(deg, n, chunk_sums);
  scan_sums_kernel<<<1, 256, 0, st>>>(chunk_sums, nchunks, rowptr + n);      // rowptr[N] = number of valid edges
  if (nchunks > 0) scan_apply_kernel<<<nchunks, 256, 0, st>>>(deg, chunk_sums, n, rowptr);
  if (e > 0) {
    csr_fill_kernel<<<(e + 255) / 256, 256, 0, st>>>(edge_index + E, e, n, rowptr, cursor, eid_tmp);
    csr_rank_kernel<<<(e + 255) / 256, 256, 0, st>>>(edge_index, e, n, rowptr, eid_tmp, eid, src, dst);
  }
  return check_launch();
}

extern "C" int isg_graph_edge_ptr(const int32_t *ptr, const int32_t *rowptr, int64_t B, int32_t *eptr, int32_t *emax,
                                  void *stream) {
  if (B < 0 || !ptr || !rowptr || !eptr) return ISG_EINVAL;
  if (B >= (1ll << 31)) return ISG_EUNSUPPORTED;
  hipStream_t st = as_stream(stream);
  if (emax) fill_i32_kernel<<<1, 64, 0, st>>>(emax, 1, 0);
  graph_eptr_kernel<<<(unsigned)((B + 1 + 255) / 256), 256, 0, st>>>(ptr, rowptr, (int)B, eptr, emax);
  return check_launch();
}


// ptr / nmax, CSR by destination, per-graph slot ranges and the largest edge count in six launches (the tile plan is the
// seventh): what isg_graph_ptr + isg_csr_build + isg_graph_edge_ptr compute in fourteen.  bounds int32[2] receives {largest node
// count, largest edge count of a graph}; the other outputs as in those three; workspace: isg_csr_workspace_bytes(N, E).
extern "C" int isg_graph_plan_build(const int64_t *batch, const int64_t *edge_index, int64_t N, int64_t E, int64_t B,
                                    int32_t *ptr, int32_t *bounds, int32_t *bounds_host, int32_t *rowptr, int32_t *eid,
                                    int32_t *src, int32_t *dst, int32_t *eptr, void *workspace, size_t workspace_bytes,
                                    void *stream) {
  if (N < 0 || E < 0 || B < 0 || !ptr || !bounds || !rowptr || !eptr || (N > 0 && !batch) || (E > 0 && (!edge_index || !eid || !src)))
    return ISG_EINVAL;
  if (N >= (1ll << 31) - 1024 || E >= (1ll << 31) - 1024 || B >= (1ll << 31) - 1024) return ISG_EUNSUPPORTED;
  if (!workspace || workspace_bytes < isg_csr_workspace_bytes(N, E)) return ISG_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  int *deg = (int *)workspace;            // N+1
  int *cursor = deg + (N + 1);            // N+1
  int *eid_tmp = cursor + (N + 1);        // E
  int *chunk_sums = eid_tmp + E;          // ceil(N / SCAN_CHUNK) + 1
  const int n = (int)N, e = (int)E, b = (int)B;
  static const bool small_ok = [] { const char *f = getenv("ISG_PLAN_SMALL"); return !f || atoi(f) != 0; }();
  if (small_ok && n <= PS_MAX_N && e <= PS_MAX_E && b <= PS_MAX_N && e > 0) {      // a handful of graphs: one workgroup, one launch
    plan_small_kernel<<<1, PS_THREADS, 0, st>>>(batch, edge_index, n, e, b, ptr, bounds, bounds_host, rowptr, eid, src, dst, eptr, deg,
                                               cursor, eid_tmp);
    return check_launch();
  }
  const int nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  const long long span1 = std::max<long long>(std::max<long long>(2ll * (n + 1), n), b + 1);
  plan_init_kernel<<<(unsigned)((span1 + 255) / 256), 256, 0, st>>>(batch, n, b, ptr, deg, 2 * (n + 1), bounds);
  const int span2 = std::max(e, b);
  if (span2 > 0) plan_hist_kernel<<<(span2 + 255) / 256, 256, 0, st>>>(edge_index + E, e, n, deg, ptr, b, bounds);
  if (nchunks > 0) {
    scan_chunk_sums_kernel<<<nchunks, 256, 0, st>>>(deg, n, chunk_sums);
    scan_apply_self_kernel<<<nchunks, 256, 0, st>>>(deg, chunk_sums, n, nchunks, rowptr);
  } else {
    fill_i32_kernel<<<1, 64, 0, st>>>(rowptr, 1, 0);
  }
  const int span3 = std::max(e, b + 1);
  plan_fill_kernel<<<(span3 + 255) / 256, 256, 0, st>>>(edge_index + E, e, n, rowptr, cursor, eid_tmp, ptr, b, eptr, bounds + 1);
  if (e > 0)
    csr_rank_kernel<<<(e + 255) / 256, 256, 0, st>>>(edge_index, e, n, rowptr, eid_tmp, eid, src, dst, bounds, bounds_host);
  else if (bounds_host)
    copy_bounds_kernel<<<1, 64, 0, st>>>(bounds, bounds_host);
  return check_launch();
}
