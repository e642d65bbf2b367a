// Minimal form of isg_gatv2_tile_conv's aggregation loop as it was while it returned different bits from launch to launch
// (DESIGN.md 15.4): a half-wave per destination node, the node's in-edge slots [rb, re) walked with `#pragma unroll 2`:
// two weights by one ds_read2_b32, two x_l rows by ds_read_b128, four v_pk_fma_f32 into one accumulator pair of which the
// second two take their weight from the high dword (op_sel:[0,1,0]).  The same sums are formed by a rolled loop; the kernel
// counts the lanes whose two results differ.  hipcc -O3 --offload-arch=gfx950 pk_fma_pair.hip -o pk_fma_pair && ./pk_fma_pair
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ROWS = 64, LDX = 132, ECAP = 256;

__global__ __launch_bounds__(256, 2) void probe(const float *__restrict__ xl, const int *__restrict__ rowptr, const int *__restrict__ src,
                                                const float *__restrict__ w, int ntiles, int rounds, unsigned long long *bad,
                                                float *sink, int mfma_team, int inflight, float *outbuf) {
  __shared__ __attribute__((aligned(16))) float sXl[ROWS][LDX];
  __shared__ __attribute__((aligned(16))) int4 s_tab[ECAP];
  __shared__ float s_w[ECAP];
  __shared__ int s_rp[ROWS + 4];
  const int tid = threadIdx.x, lane = tid & 63, tw = tid >> 6, fr = lane & 31, hh = lane >> 5;
  unsigned long long mine = 0;
  float keep = 0.f;
  if (mfma_team && blockIdx.x >= gridDim.x / 2) {
    // the other workgroup of the CU (blocks g and g + grid / 2 land on the same CU under round-robin dispatch): matrix-core work on
    // the SIMDs the probe's packed operations run on, as the second (tile, head) workgroup of isg_gatv2_tile_conv does
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (tid - i)); }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int it = 0; it < mfma_team; ++it) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
    }
    if (acc0[0] + acc1[3] == 123.456f) sink[1] = acc0[0];
    return;
  }
  const int nprobe = mfma_team ? gridDim.x / 2 : gridDim.x;
  for (int t = blockIdx.x; t < ntiles; t += nprobe) {
    __syncthreads();
    for (int i = tid; i < ROWS * 32; i += 256) {
      const int r = i >> 5, c = i & 31;
      *reinterpret_cast<float4 *>(&sXl[r][c * 4]) = *reinterpret_cast<const float4 *>(xl + ((size_t)t * ROWS + r) * 128 + c * 4);
    }
    if (tid <= ROWS) s_rp[tid] = rowptr[t * (ROWS + 1) + tid];
    s_tab[tid] = make_int4(src[t * ECAP + tid], 0, 0, 0);
    s_w[tid] = w[t * ECAP + tid];
    __syncthreads();
    const int ne = s_rp[ROWS];
    for (int round = 0; round < rounds; ++round) {
      // (isg_gatv2_tile_conv has the NEXT tile's x_l rows on their way into registers while it aggregates: eight 16-byte
      // non-temporal loads per lane, issued here and used behind the loop)
      typedef float nt4 __attribute__((ext_vector_type(4)));
      nt4 pre[8];
      if (inflight) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          pre[u] = __builtin_nontemporal_load(reinterpret_cast<const nt4 *>(
              xl + ((size_t)((t + 1 + round) % ntiles) * ROWS + (tid >> 5) + 8 * u) * 128 + (tid & 31) * 4));
      }
#pragma unroll 1
      for (int k = 2 * tw + hh; k < ROWS; k += 8) {
        const int rb = s_rp[k], re = min(s_rp[k + 1], ne);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
        for (int s = rb; s < re; ++s) {
          const float wm = s_w[s];
          const float4 u4 = *reinterpret_cast<const float4 *>(&sXl[s_tab[s].x][fr * 4]);
          o.x = fmaf(u4.x, wm, o.x);
          o.y = fmaf(u4.y, wm, o.y);
          o.z = fmaf(u4.z, wm, o.z);
          o.w = fmaf(u4.w, wm, o.w);
        }
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int s = rb; s < re; ++s) {
          const float wm = s_w[s];
          const float4 u4 = *reinterpret_cast<const float4 *>(&sXl[s_tab[s].x][fr * 4]);
          p.x = fmaf(u4.x, wm, p.x);
          p.y = fmaf(u4.y, wm, p.y);
          p.z = fmaf(u4.z, wm, p.z);
          p.w = fmaf(u4.w, wm, p.w);
        }
        if (o.x != p.x) mine += 1ull;
        if (o.y != p.y) mine += 1ull << 16;
        if (o.z != p.z) mine += 1ull << 32;
        if (o.w != p.w) mine += 1ull << 48;
        keep += o.x + o.y + o.z + o.w;
        if (inflight) {        // the aggregated row leaves as it does in the kernel: a non-temporal 16-byte store per lane
          nt4 o4 = {o.x, o.y, o.z, o.w};
          __builtin_nontemporal_store(o4, reinterpret_cast<nt4 *>(outbuf + ((size_t)blockIdx.x * ROWS + k) * 128 + fr * 4));
        }
      }
      if (inflight) {
#pragma unroll
        for (int u = 0; u < 8; ++u) keep += pre[u][0] + pre[u][3];
      }
    }
  }
  if (mine) atomicAdd(bad, mine);
  if (keep == 123.456f) sink[0] = keep;
}

int main(int argc, char **argv) {
  const int ntiles = 2048, rounds = argc > 1 ? atoi(argv[1]) : 20, launches = argc > 2 ? atoi(argv[2]) : 20;
  const int inflight = argc > 4 ? atoi(argv[4]) : 0;      // 1: loads in flight across the loop and stores inside it, as in the kernel
  float *d_out;
  hipMalloc(&d_out, (size_t)512 * ROWS * 128 * 4);
  const int mfma_team = argc > 3 ? atoi(argv[3]) : 0;      // > 0: half of the workgroups run that many MFMA pairs beside the probe
  float *h_xl = (float *)malloc((size_t)ntiles * ROWS * 128 * 4), *h_w = (float *)malloc((size_t)ntiles * ECAP * 4);
  int *h_rp = (int *)malloc((size_t)ntiles * (ROWS + 1) * 4), *h_src = (int *)malloc((size_t)ntiles * ECAP * 4);
  srand(7);
  for (size_t i = 0; i < (size_t)ntiles * ROWS * 128; ++i) h_xl[i] = (float)(rand() % 2001 - 1000) / 257.f;
  for (int t = 0; t < ntiles; ++t) {
    int at = 0;
    for (int k = 0; k < ROWS; ++k) {       // in-degrees 1..6 like the scene graphs (self loop + ~1.5 edges), some rows empty
      h_rp[t * (ROWS + 1) + k] = at;
      at += k < 54 ? 1 + rand() % 6 : 0;
      if (at > ECAP) at = ECAP;
    }
    h_rp[t * (ROWS + 1) + ROWS] = at;
    for (int s = 0; s < ECAP; ++s) {
      h_src[t * ECAP + s] = rand() % 54;
      h_w[t * ECAP + s] = (float)(rand() % 1000) / 1000.f;
    }
  }
  float *d_xl, *d_w, *d_sink;
  int *d_rp, *d_src;
  unsigned long long *d_bad, h_bad = 0;
  hipMalloc(&d_xl, (size_t)ntiles * ROWS * 128 * 4); hipMalloc(&d_w, (size_t)ntiles * ECAP * 4); hipMalloc(&d_sink, 8);
  hipMalloc(&d_rp, (size_t)ntiles * (ROWS + 1) * 4); hipMalloc(&d_src, (size_t)ntiles * ECAP * 4); hipMalloc(&d_bad, 8);
  hipMemcpy(d_xl, h_xl, (size_t)ntiles * ROWS * 128 * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_w, h_w, (size_t)ntiles * ECAP * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_rp, h_rp, (size_t)ntiles * (ROWS + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_src, h_src, (size_t)ntiles * ECAP * 4, hipMemcpyHostToDevice);
  unsigned long long tot[4] = {0, 0, 0, 0};
  for (int l = 0; l < launches; ++l) {
    hipMemset(d_bad, 0, 8);
    probe<<<512, 256>>>(d_xl, d_rp, d_src, d_w, ntiles, rounds, d_bad, d_sink, mfma_team, inflight, d_out);
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost);
    for (int c = 0; c < 4; ++c) tot[c] += (h_bad >> (16 * c)) & 0xffff;
  }
  const double sums = (double)launches * ntiles * rounds * 54.0 * 32.0;
  printf("mfma team %d, loads in flight %d: ", mfma_team, inflight);
  printf("lanes whose unrolled and rolled sums differ, by component x y z w: %llu %llu %llu %llu  (of %.3g lane sums per component)\n",
         tot[0], tot[1], tot[2], tot[3], sums);
  return (tot[0] | tot[1] | tot[2] | tot[3]) ? 1 : 0;
}
