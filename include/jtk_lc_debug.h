/* jtk_lc_debug.h -- diagnostic entry points of libjtk_lc.so.  NOT part of the drop-in boundary (include/jtk_lc.h): nothing a
 * jtk host needs is declared here.  They exist so that a test can look at an intermediate the boundary does not return
 * (tests/test_gpu_correction.py compares the device-filled similarity matrix of phmm_likelihood_correction.rs:272-285 bit for
 * bit with the oracle's).  Same library, same build: there is no separate test build of the product. */
#ifndef JTK_LC_DEBUG_H
#define JTK_LC_DEBUG_H

#include "jtk_lc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Keep (on != 0) the raw similarity matrix of the first corrected chunk of the following jtk_lc_correct_clustering calls on
 * this thread. */
JTK_LC_API void jtk_lc_debug_cc_keep_sims(int on);
/* Copy up to `cap` doubles of the kept matrix into `out`; returns its size. */
JTK_LC_API size_t jtk_lc_debug_cc_first_sims(double *out, size_t cap);

/* Where the chain kernel's time went, per chunk of a session that has run: cycles[c] = shader-clock cycles the chunk's consumer
 * wave spent in the kernel (0 for a chunk without a variant column), events[c] = proposals of its table-driven chains that could
 * not be stepped over (accepted moves, rounding residues, draws inside the guard bands).  A chain launch lasts as long as its
 * slowest chunk: this is how bench.py names it.  Either pointer may be null. */
JTK_LC_API int jtk_lc_debug_chain_profile(jtk_lc_session_t *s, uint64_t *cycles, uint32_t *events);

#ifdef __cplusplus
}
#endif
#endif /* JTK_LC_DEBUG_H */
