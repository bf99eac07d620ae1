// sweep2_fam.hip -- one piece of sweep2.hip's template instantiations (see sweep2_kernels.h): compiled six times by the Makefile,
//   -DS2_FAM=FAM_SE|FAM_M3|FAM_M5  -DS2_PIECE=0 (full sweeps: sweep2_kernel with and without the moving alpha window)
//                                            1 (kept candidate state: sweep2_kernel<.., PART> and sweep2_rank1_kernel)
#include "sweep2_kernels.h"
#if S2_PIECE == 0
template int launch_s2_fam<S2_FAM>(const SweepArgs &, int64_t, hipStream_t);
#else
template int launch_s2_rank1_fam<S2_FAM>(const SweepArgs &, int64_t, hipStream_t);
template int launch_s2_part_fam<S2_FAM>(const SweepArgs &, int64_t, hipStream_t);
#endif

// (see small2.hip: ibo_touch_small2) one kernel of this piece
#define S2_TOUCH_NAME2(f, p) ibo_touch_s2fam_##f##_##p
#define S2_TOUCH_NAME(f, p) S2_TOUCH_NAME2(f, p)
void S2_TOUCH_NAME(S2_FAM, S2_PIECE)()
{
    hipFuncAttributes a;
#if S2_PIECE == 0
    (void)hipFuncGetAttributes(&a, (const void *)sweep2_kernel<S2_FAM, 2, false>);
#else
    (void)hipFuncGetAttributes(&a, (const void *)sweep2_rank1_kernel<S2_FAM, 2>);
#endif
}
