// Diagnostic builds only.  Everything the kernels need to be timed from the inside sits behind ONE switch, -DISG_DIAG, and in
// this one header; a production build (no ISG_DIAG) compiles every macro below to nothing, and the kernels carry one line per
// stamp and no #if of their own.
//
// Stamps: every wave reads the core clock (s_memtime) at the boundaries of its segments and adds (persistent kernels) or sets
// (one pass per launch) the differences in 16 private counters; lane 0 writes them to a buffer the tool handed in through the
// kernel's setter, [workgroup or tile][wave][16] int64.  No output value depends on a stamp, but a stamp is a scheduling
// fence: a stamped build says where a wave waits, not what an un-stamped build costs (profiles/r04_z_h3p_piece_placement_ab.txt
// has a case where the two disagree).  Tools: tools/stamp_layer_conv.py, stamp_dense_tail.py, stamp_tile_conv.py and
// stamp_h3p.py build their own library with -DISG_DIAG.
//
// The compile-time ABLATION builds of rounds 3 and 4 (LC_ABL / DT_ABL bit masks in the layer kernel and the dense tail, the
// run-time `abl` field of the planes32 GEMM, the DBG arms and stamps of the round-2 edge-logits kernel) are gone from the
// sources: their tables are profiles/r03_bg_layer_conv_ablation.md,
// profiles/r03_bh_dense_tail_ablation.md, profiles/r04_b_h3p_ablation.txt and profiles/r02_w_edge_logits.md, the hooks are in
// the history (round-4 commit
// "Engine: a finished piece's arithmetic inside the MFMA segment ..." is the last one that has them).
#ifndef ISG_DIAG_HPP
#define ISG_DIAG_HPP

// -DISG_DIAG_STRICT (a second switch, independent of the stamps): the SCHEDULE-FREE build of the hand-scheduled kernels.  Every
// hand-counted wait ("all but the n youngest requests have landed") becomes vmcnt(0) lgkmcnt(0), every raw s_barrier a full
// __syncthreads() (fences and the compiler's own waits included).  Arithmetic is untouched, so the strict build must return the
// bits of the fast build on every input: when the two differ the fault is in a wait count or a missing barrier, not in the
// arithmetic.  build() makes csrc/libisg_hip_strict.so beside the library; tests/test_gpu_strict.py compares the two.
#ifdef ISG_DIAG_STRICT
#define ISG_WAIT(imm) ((void)(imm), __builtin_amdgcn_s_waitcnt(0))
#define ISG_BARRIER() __syncthreads()
#else
#define ISG_WAIT(imm) __builtin_amdgcn_s_waitcnt(imm)
#define ISG_BARRIER() __builtin_amdgcn_s_barrier()
#endif

#ifdef ISG_DIAG
#define ISG_DIAG_BUFFER(name) static __device__ long long *name = nullptr;
#define ISG_DIAG_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define ISG_DIAG_BEGIN()                                                                                           \
  long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                         \
  const long long st_begin = ISG_DIAG_NOW();                                                                       \
  long long st_last = st_begin;
#define ISG_DIAG_ADD(i) { const long long now_ = ISG_DIAG_NOW(); st_acc[i] += now_ - st_last; st_last = now_; }
#define ISG_DIAG_SET(i) { const long long now_ = ISG_DIAG_NOW(); st_acc[i] = now_ - st_last; st_last = now_; }
// the accumulators a segment produced are "used" here, so that its MFMAs cannot sink below the stamp that closes it
#define ISG_DIAG_KEEP2(x, y) asm volatile("" ::"v"(x), "v"(y));
// lane 0 of a wave writes its counters to row `row` of `buf`; counter `total` = the wave's lifetime; `extra`: more assignments
#define ISG_DIAG_DUMP(buf, row, total, extra)                                                                      \
  if (buf && lane == 0) {                                                                                          \
    st_acc[total] = ISG_DIAG_NOW() - st_begin;                                                                     \
    extra                                                                                                          \
    long long *dst_ = buf + (long long)(row) * 16;                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) dst_[i_] = st_acc[i_];                                       \
  }
#define ISG_DIAG_SETTER(fn, buf)                                                                                   \
  extern "C" int fn(long long *p) { return hipMemcpyToSymbol(HIP_SYMBOL(buf), &p, sizeof(p)) == hipSuccess ? 0 : -1; }
#else
#define ISG_DIAG_BUFFER(name)
#define ISG_DIAG_BEGIN()
#define ISG_DIAG_ADD(i)
#define ISG_DIAG_SET(i)
#define ISG_DIAG_KEEP2(x, y)
#define ISG_DIAG_DUMP(buf, row, total, extra)
#define ISG_DIAG_SETTER(fn, buf)
#endif

#endif  // ISG_DIAG_HPP
