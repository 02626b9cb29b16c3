// kernels.h -- gfx950 (CDNA4) device kernels of libbqhip.so, by topic.
//
// Everything is fp64 and column-major.  All kernels take a batch dimension in
// blockIdx.z (independent problems / hyper-parameter points) with element
// strides, so one launch covers a whole shard of problems.  DESIGN.md section 4 has
// the roofline that bounds each kernel and its algorithmic work.
#pragma once
#include "common.h"
#include "gram.h"    // gram_sym_kernel, gram_cross_kernel, assemble_kernel
#include "potf2.h"   // potf2_kernel, potf2f_body (the 4-wave diagonal factor, fusable into the step kernels)
#include "trsm.h"    // trsm_blk_kernel, diag_winv_kernel
#include "trsv.h"    // trsv_diag/fwd/bwd_kernel: single right-hand-side sweeps (GEMV form)
#include "gemm.h"    // gemm_sub_kernel, gemm_k64_kernel, gemm_lds_kernel
#include "slab.h"    // slab_step_kernel: one launch per 64-column step of a small system
#include "reduce.h"  // finalize_kernel, rowdot_kernel, predict_mean_kernel, logdet_kernel, ...
#include "probe.h"   // probe_* kernels
#include "moments.h" // int_K*, iikk, esm kernels
