// Timing ablations of the GEMM kernels (gemm_f32.hip): each ABL_* switch removes one ingredient of the K loop --
// the operand split, the LDS stores, the global loads, the barrier ... -- to price it (profiles/r01_gemm_ablation*.log,
// profiles/r02_gemm_x3_ablations.txt).  Several of them produce WRONG numbers by design.  They are not part of the
// product: unless the translation unit is built with -DPLNLP_ABLATION (scripts/build_x3_ablations.sh,
// scripts/ablate/run.sh do), every switch is forced off here, whatever the command line says.
#pragma once
#ifndef PLNLP_ABLATION
#undef ABL_X3_NOSPLIT
#undef ABL_X3_NOSTORE
#undef ABL_X3_NOGLOAD
#undef ABL_X3_PLAIN_LOOP
#undef ABL_NOSTAGE
#undef ABL_NOSTORE
#undef ABL_NOGLOAD
#undef ABL_NOBARRIER
#undef ABL_CLAMPED_LOOP
#endif
