/* oracle/levmar_nolapack.h -- build recipe only (no reference source is copied): force-included in front of the reference's
 * levmar-2.6 sources (gcc -include) so that they compile in the configuration levmar itself documents for systems without
 * LAPACK -- its own LU (Ax_eq_b_LU_noLapack, external/levmar-2.6/Axb_core.c:1123-1277) instead of dgetrf / dgetrs.
 * levmar.h is included here first (its include guard then keeps the later #include "levmar.h" of lm.c / Axb.c / misc.c from
 * defining HAVE_LAPACK again) and the macro is removed.  Result: oracle/_ref/liblevmar_nolapack_ref.so, the reference's
 * dlevmar_dif built from its own files with NO external library and NO stand-in. */
#include "levmar.h"
#undef HAVE_LAPACK
