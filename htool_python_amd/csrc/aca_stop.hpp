// aca_stop.hpp -- the stopping rule of the partially pivoted ACA, shared by every implementation of it in the library
// (host_aca for callback generators, aca_wave / aca_multi / aca_kernel / the lockstep kernels for native ones).
//
// Reference behaviour: stop as soon as |u_k| |v_k| <= eps |A_k|_F.  That estimate looks at ONE rank-one term;
// on degenerate clouds (nearly collinear points: the `sheet` cases of tools/fuzz.py) partial pivoting can pass it while a
// large part of the block has not been seen yet -- profiles/r03_fuzz_sheet_case_61_322.txt: 65 eps left at rank 3, where
// one more step finds a term a thousand times larger and the iteration ends at rank 6 with 0.06 eps.
// htool_build_params.aca_confirm_steps = c > 0 asks for c further steps that pass the test as well before the iteration stops;
// the confirming terms are then DROPPED (the leaf keeps the rank at which the test first passed), so on geometries where
// the estimate was right all along the factors are bit for bit those of c = 0 -- the safeguard costs time (one more pivot
// step per leaf and unit of c), not memory, and changes a leaf only when a confirming step fails.
#pragma once
#if defined(__HIPCC__)
#define HM_HD __host__ __device__ __forceinline__
#else
#define HM_HD inline
#endif

namespace hm {

// the kernels take ONE integer for "reqrank": r >= 0 = exactly r steps (no test), -1 - c = test with c confirming steps
HM_HD int aca_reqrank_argument(int reqrank, int confirm) { return reqrank >= 0 ? reqrank : -1 - (confirm > 0 ? confirm : 0); }
HM_HD int aca_confirm_steps(int reqrank_argument) { return reqrank_argument < 0 ? -1 - reqrank_argument : 0; }

struct AcaStop {
    // ONE integer of state (the register ACA kernels run at their register limit): bits 20.. = consecutive steps, up to the
    // last one, that passed the test; bits 0..19 = the rank after the first of them.  0 = no passed test pending.
    int st = 0;
    // after step number k (k terms computed): 0 = go on, 1 = stop and keep k terms (k may have been reset to the rank of the
    // first pass), 2 = stop, the block is not worth storing in low-rank form.  Same order of checks as the reference loop: size first.
    HM_HD int after_step(int &k, bool passed, bool too_big, bool no_next_row, int confirm) {
#ifdef HTOOL_ACA_NO_CONFIRM // (A/B builds: the reference's rule only, no state)
        (void)confirm;
        if (too_big) return 2;
        return (passed || no_next_row) ? 1 : 0;
#else
        const int before = st;
        st = passed ? (before != 0 ? before + (1 << 20) : (k | (1 << 20))) : 0;
        // too many terms to be worth storing.  A pass that was waiting for its confirmation is accepted only when THIS step passed
        // as well: a confirming step that fails has just shown the earlier rank to be premature (round 3 accepted it anyway and
        // defeated the safeguard exactly where it matters); the block then goes the way of every block that is not compressible
        if (too_big) { if (before != 0 && passed) { k = before & 0xfffff; return 1; } return 2; }
        if (passed && (st >> 20) > confirm) { k = st & 0xfffff; return 1; }
        if (no_next_row) { if (st != 0) k = st & 0xfffff; return 1; }
        return 0;
#endif
    }
    // the iteration ends for another reason (capacity reached, every row used, rank limit): true when a passed test is pending,
    // in which case k is reset to the rank it was passed at and the leaf is accepted
    HM_HD bool settle(int &k) {
        if (st != 0) { k = st & 0xfffff; return true; }
        return false;
    }
};

} // namespace hm
