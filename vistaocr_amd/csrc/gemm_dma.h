// Host interface of the DMA-staged panel GEMM (gemm_dma.hip) for gemm.hip's entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace vocr_dma_gemm {

struct Plan { bool ok; int panels, groups, ksplit, kps; size_t slab_bytes; };

// nprob independent products of one shape, or nseg K segments summed into one C (one of the two is 1).  `no_split`: the caller cannot
// run the slab reduce (never the case today).  ok == false: the shape is left to gemm.hip's tile kernel.
Plan plan(int transa, int transb, int m, int n, int k, int lda, int ldb, int ldc, int nprob, int nseg, bool no_split);

// Launch (and nothing else: when p.ksplit > 1 the caller adds the slabs [tile][split][256][128] with splitk_reduce_kernel).
int launch(const Plan& p, int transa, int transb, int m, int n, int k, const float* const a[2], int lda, const float* const b[2], int ldb,
           float* const c[2], int ldc, const float* const bias[2], int relu, int nprob, int nseg, float* slab, hipStream_t s);

}  // namespace vocr_dma_gemm
