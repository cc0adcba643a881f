// The recurrences of the QP subproblem that are serial in exact arithmetic, for ONE wavefront with its operands in
// registers (device only; n <= 128).  lsq_spmd.hpp calls them from the team's first wavefront while the other
// wavefronts wait at the next barrier.  Every entry is computed with the operations and in the order of the host
// core (slsqp_core.c), like everything else in the team code: same bits.
//
// Why: a recurrence of length n whose steps meet at a workgroup barrier costs ~700 cycles per step (LDS round trips
// and the barrier), and an ordered sum whose terms one thread fetches from LDS ~30 cycles per term.  A wavefront that
// keeps the vector in registers and passes single values between lanes through the scalar unit (v_readlane) or the
// DPP row broadcast pays the FP64 dependent-issue latency instead (~5 cycles per operation).
//
// What the steps of such a chain may contain was measured (tools/micro/*.hip, one wavefront on an idle CU, cycles
// added to a dependent multiply-add of 10): v_readlane of a double into the next operation +28; a select whose mask
// a v_cmp in the same step wrote +36, with the mask computed by the scalar unit +16; a branch on a value of the chain
// +52; a plain FP64 division ~150.  Hence: masks come from the scalar unit (select_lanes), divisions are split
// (recip_of / quotient) and nothing in a chain branches on its values (see Flags).
#pragma once

#if defined(__HIP_DEVICE_COMPILE__)

namespace lsqwave {

constexpr int kMaxN = 128;

// The lane number, opaque to the optimiser: what these routines derive from it (slot numbers, clamped indices) is then
// computed where it is used.  Hoisted out of the kernel's loop over problems it would stay in registers through the
// row transform, which has none to spare.
__device__ __forceinline__ int lane_here()
{
    int lane = (int)(threadIdx.x & 63u);
    asm volatile("" : "+v"(lane));
    return lane;
}

__device__ __forceinline__ double bcast(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---- selects whose mask the scalar unit computes ------------------------------------------------------------------
// lanes whose bit is set in `lanes` take `take`, the others keep `keep`
__device__ __forceinline__ double select_lanes(double keep, double take, unsigned long long lanes)
{
    int lo = __double2loint(keep), hi = __double2hiint(keep);
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lo) : "v"(__double2loint(take)), "s"(lanes));
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(hi) : "v"(__double2hiint(take)), "s"(lanes));
    return __hiloint2double(hi, lo);
}
// the lanes above `lane` (none for lane >= 63)
__device__ __forceinline__ unsigned long long lanes_above(int lane) { return lane >= 63 ? 0ull : ~0ull << (lane + 1); }

// ---- a division whose divisor is known early -------------------------------------------------------------------
// num / den as the compiler's own expansion computes it (v_div_scale, v_rcp and two Newton steps on the divisor, then
// q = num y, r = fma(-den, q, num), fma(r, y, q), v_div_fixup), split in two: everything that depends on the divisor
// alone (recip_of) can be done long before the dividend exists, and the three operations that are left are all a
// recurrence has to wait for.  v_div_scale leaves its operands alone and v_div_fixup passes the quotient through
// unless an exponent is extreme: inside the band 2^-400 <= |v| <= 2^400 for both operands the three operations give
// the bits of num / den (tools/micro/div_split.hip: 2^30 random pairs).  Outside it (zeros, infinities and NaN
// included) they may not, and a chain must not branch on its values: every operand is recorded in Flags instead, and
// a routine whose flags came up returns false, having restored its input: the caller runs the team code instead.
struct Flags {
    int bits = 0;                                         // sign bit set: some operand was outside the band
    __device__ __forceinline__ void see(double v)
    {
        const int e = (int)(((unsigned)__double2hiint(v) >> 20) & 0x7ffu);
        bits |= (e - 623) | (1423 - e);
    }
    __device__ __forceinline__ bool clean() const { return bits >= 0; }
};
__device__ __forceinline__ double recip_of(double den, Flags &fl)
{
    fl.see(den);
    double y = __builtin_amdgcn_rcp(den);
    double e = fma(-den, y, 1.0);
    y = fma(y, e, y);
    e = fma(-den, y, 1.0);
    y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double quotient(double num, double den, double y, Flags &fl)
{
    fl.see(num);
    const double q = num * y;
    const double r = fma(-den, q, num);
    return fma(r, y, q);
}
// does any lane of the wavefront hold a raised flag?
__device__ __forceinline__ bool all_clean(const Flags &fl) { return __builtin_amdgcn_ballot_w64(fl.bits < 0) == 0ull; }

// ---- an ordered sum at the FP64 pipeline's latency ----------------------------------------------------------------
// 128 slots held 16 lanes x 8 registers, every row of 16 lanes a copy: slot s = 16 r + c is p[r] of the lanes with
// lane % 16 == c.  Returns ((t + slot(first)) + slot(first + 1)) + ... + slot(127) in every lane.
// One instruction per term: v_fmac_f64 with its first factor taken from lane c of the row (DPP row_newbcast) and the
// second 1.0, i.e. t = p * 1.0 + t.  The product is exact, so the one rounding is the addition's: the bits of t + p.
// The 128 instructions are laid out back to back (8 bytes each) and entered at the first term by a computed jump.
// The jump's arithmetic relies on two encodings - 12 bytes of scalar instructions between the address s_getpc returns
// and the first term, 8 bytes per term - and the assembler checks both where the block is emitted (.if / .error on
// the label differences); tests/test_gpu_solve.py::test_ordered_sum_enters_the_chain_at_every_term drives every entry.
#define LSQW_T(r, c) "v_fmac_f64_dpp %[t], %[p" #r "], %[one] row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define LSQW_R(r)                                                                                                  \
    LSQW_T(r, 0) LSQW_T(r, 1) LSQW_T(r, 2) LSQW_T(r, 3) LSQW_T(r, 4) LSQW_T(r, 5) LSQW_T(r, 6) LSQW_T(r, 7)          \
    LSQW_T(r, 8) LSQW_T(r, 9) LSQW_T(r, 10) LSQW_T(r, 11) LSQW_T(r, 12) LSQW_T(r, 13) LSQW_T(r, 14) LSQW_T(r, 15)
__device__ __forceinline__ double ordered_sum_from(double t, const double (&p)[8], int first)
{
    const double one = 1.0;
    // the jump lands (s_getpc_b64 gives the address of the instruction after it) 12 bytes of scalar instructions
    // plus 8 bytes per skipped term further on; first == 128 skips them all
    // (first is the same in every lane; said explicitly, because where the compiler cannot prove it - inside a function
    //  that is not inlined, whose arguments arrive in vector registers - the "s" operand below would be handed one)
    const int off = 12 + 8 * __builtin_amdgcn_readfirstlane(first);
    asm volatile("s_nop 1\n\t"                       // a VALU result (p) may not be read through DPP in the next two slots
                 "s_getpc_b64 s[38:39]\n"
                 ".Llsqw_pc%=:\n\t"
                 "s_add_u32 s38, s38, %[off]\n\t"
                 "s_addc_u32 s39, s39, 0\n\t"
                 "s_setpc_b64 s[38:39]\n"
                 ".Llsqw_first%=:\n\t"
                 LSQW_R(0) LSQW_R(1) LSQW_R(2) LSQW_R(3) LSQW_R(4) LSQW_R(5) LSQW_R(6) LSQW_R(7)
                 "\n.Llsqw_end%=:\n\t"
                 ".if (.Llsqw_first%= - .Llsqw_pc%=) != 12\n\t"
                 ".error \"ordered_sum_from: the scalar instructions of the computed jump are not 12 bytes\"\n\t"
                 ".endif\n\t"
                 ".if (.Llsqw_end%= - .Llsqw_first%=) != 1024\n\t"
                 ".error \"ordered_sum_from: a term is not one 8-byte instruction\"\n\t"
                 ".endif\n\t"
                 : [t] "+v"(t)
                 : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]),
                   [p6] "v"(p[6]), [p7] "v"(p[7]), [one] "v"(one), [off] "s"(off)
                 : "s38", "s39", "scc");
    return t;
}
#undef LSQW_R
#undef LSQW_T

#define LSQW_EO(r) ((r) * n - ((r) * ((r) - 1)) / 2)

// what one wavefront wrote to LDS, it reads back afterwards: no other wavefront is waited for
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// x = e^-1 x by back substitution (e upper triangular, packed by rows): x(i) = (x(i) - sum_{j > i} e(i, j) x(j)) / e(i, i)
// with the sum in ascending j from 0.0, rows in descending i.  x(j) becomes known in the opposite order, so the
// additions of all rows form one chain of n(n-1)/2 dependent additions: ordered_sum_from runs it at ~8 cycles per term.
// Index j sits in slot j + 128 - n, so that every row's sum ends at the last slot.
// All 64 lanes of the wavefront call it; the lanes compute the same x redundantly, lane 0 stores it.
// scratch [2n].  Returns false, with x as it was, if an operand of a division was outside the band (see Flags).
struct BackSub {
    const double *er;      // er[j] = e(i, j) for the current row i; row i - 1 starts n - i entries earlier
    const double *ys;      // the diagonal's reciprocals (recip_of)
    double *x;
    int n, pad, lane, c;
    int jc[8];
    Flags fl;
    // x by slot.  A vector type, not an array: the one entry that a row finishes is written with a register index
    // (s_set_gpr_idx), and an array that is indexed dynamically would live in scratch memory.
    typedef double Slots __attribute__((ext_vector_type(8)));
    Slots X;

    // Row i: its entries are in `cur` (fetched while the row before was summed), the next row's go to `next`.
    __device__ __forceinline__ void row(int i, const double (&cur)[8], double (&next)[8])
    {
        const double diag = er[i], xi0 = x[i], yi = ys[i];
        const double *en = i > 0 ? er - (n - i) : er;
#pragma unroll
        for (int r = 0; r < 8; ++r) next[r] = en[jc[r]];
        double P[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) P[r] = cur[r] * X[r];
        const double t = ordered_sum_from(0.0, P, i + 1 + pad);
        const double xi = quotient(xi0 - t, diag, yi, fl);
        if (lane == 0) x[i] = xi;
        const int slot = i + pad, rb = slot >> 4;
        const double old = X[rb];
        X[rb] = select_lanes(old, xi, 0x0001000100010001ull << (slot & 15));
        er = en;
    }
};

__device__ __forceinline__ bool back_substitute(const double *E, double *x, double *scratch, int n)
{
    BackSub B;
    double *ys = scratch, *x0 = scratch + n;
    B.x = x; B.ys = ys; B.n = n; B.pad = kMaxN - n;
    B.lane = lane_here(); B.c = B.lane & 15;
    for (int i = B.lane; i < n; i += 64) {
        ys[i] = recip_of(E[LSQW_EO(i)], B.fl);
        x0[i] = x[i];
    }
    // Slots that a row's sum never reaches (j <= i, and the 128 - n slots before index 0) may hold anything: their
    // entries are fetched unconditionally, from inside the triangle (row i's er[j] with 0 <= j < i is an entry of an
    // earlier row).
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int j = 16 * r + B.c - B.pad;
        B.jc[r] = j >= 0 ? j : 0;
        B.X[r] = j >= 0 ? x[j] : 0.0;
    }
    B.er = E + LSQW_EO(n - 1) - (n - 1);
    double ea[8], eb[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ea[r] = B.er[B.jc[r]];
    wave_sync();
    int i = n - 1;
    for (; i >= 1; i -= 2) {
        B.row(i, ea, eb);
        B.row(i - 1, eb, ea);
    }
    if (i == 0) B.row(0, ea, eb);
    if (all_clean(B.fl)) return true;
    wave_sync();
    for (int j = B.lane; j < n; j += 64) x[j] = x0[j];
    return false;
}

// f = e^-T g by forward substitution, column-oriented: once f(k) is known every later row i adds its term e(k, i) f(k),
// in the order k = 0, 1, ... of the serial dot product; f(k + 1) = (g(k + 1) - acc(k + 1)) / e(k + 1, k + 1).
// Lane l keeps the running sums of rows l and l + 64; the finished sum of row k + 1 reaches every lane by v_readlane.
// The operands of a step (two entries of row k per lane, g(k + 1), the diagonal entry and its reciprocal) are fetched
// three steps ahead into a ring of four register sets, the loop unrolled by four so that the ring is addressed
// statically.  Writes f (not negated) to ff.  All 64 lanes call it.  ys [n]: scratch.
// Returns false if an operand of a division was outside the band (see Flags): ff is then to be computed again.
struct FwdSub {
    const double *E, *g, *ys;
    double *ff;
    int n, lane, l0, l1;
    double a0, a1, fk;
    double e0[4], e1[4], gk[4], dk[4], yk[4];
    Flags fl;

    // The operands of step k into set q.  Rows that a step does not reach (i <= k, and i >= n in the second register)
    // accumulate entries of other rows, fetched unconditionally from inside the triangle; nobody reads those sums.
    // Past the last step the last one's operands are fetched again, unused: no branch.
    __device__ __forceinline__ void fetch(int k, int q)
    {
        const int kk = k < n - 2 ? k : (n >= 2 ? n - 2 : 0);
        const double *ek = E + LSQW_EO(kk) - kk;             // ek[i] = e(kk, i); ek[n] = e(kk + 1, kk + 1)
        const int k1 = n >= 2 ? kk + 1 : 0;
        e0[q] = ek[l0];
        e1[q] = ek[l1];
        gk[q] = g[k1];
        dk[q] = n >= 2 ? ek[n] : ek[0];
        yk[q] = ys[k1];
    }
    // step k: row k's terms, then f(k + 1), whose sum sits in lane (k + 1) % 64 of a0 (kLow: k + 1 < 64) or a1
    template <bool kLow>
    __device__ __forceinline__ void step(int k, int q)
    {
        if (kLow) a0 = a0 + e0[q] * fk;                   // rows below 64 are finished once k + 1 >= 64
        a1 = a1 + e1[q] * fk;
        const int o = k + 1;
        const double t = bcast(kLow ? a0 : a1, o & 63);
        fk = quotient(gk[q] - t, dk[q], yk[q], fl);
        ff[o] = fk;                                       // every lane stores the same value: no exec mask to change
    }
    // Steps k .. k + 3, those that exist; their operands are in sets 3, 0, 1, 2.  k + 1 is a multiple of four, so the
    // four sums wanted (rows k + 1 .. k + 4) are all below 64 or all from 64 up.
    template <bool kLow>
    __device__ __forceinline__ void four(int k)
    {
        fetch(k + 3, 2); if (k + 1 < n) step<kLow>(k, 3);
        fetch(k + 4, 3); if (k + 2 < n) step<kLow>(k + 1, 0);
        fetch(k + 5, 0); if (k + 3 < n) step<kLow>(k + 2, 1);
        fetch(k + 6, 1); if (k + 4 < n) step<kLow>(k + 3, 2);
    }
};

__device__ __forceinline__ bool forward_substitute(const double *E, const double *g, double *ff, double *ys, int n)
{
    FwdSub F;
    F.E = E; F.g = g; F.ys = ys; F.ff = ff; F.n = n;
    F.lane = lane_here();
    F.l0 = F.lane < n ? F.lane : n - 1;
    F.l1 = F.lane + 64 < n ? F.lane + 64 : n - 1;
    for (int i = F.lane; i < n; i += 64) ys[i] = recip_of(E[LSQW_EO(i)], F.fl);
    F.a0 = 0.0; F.a1 = 0.0;
    F.fk = (g[0] - 0.0) / E[0];
    ff[0] = F.fk;
    wave_sync();
    F.fetch(0, 0); F.fetch(1, 1); F.fetch(2, 2);
    F.fetch(3, 3); if (1 < n) F.step<true>(0, 0);                 // steps 0, 1, 2 (rows 1, 2, 3) ...
    F.fetch(4, 0); if (2 < n) F.step<true>(1, 1);
    F.fetch(5, 1); if (3 < n) F.step<true>(2, 2);
    int k = 3;                                                    // ... then row blocks 4..7, 8..11, ...
    for (; k + 1 < n && k + 1 < 64; k += 4) F.four<true>(k);
    for (; k + 1 < n; k += 4) F.four<false>(k);
    return all_clean(F.fl);
}

// v = L^-1 z for the unit lower triangular L of the packed LDL' factor a (column i at a[off(i) ..], off as above),
// column-oriented as in slsqp_core.c's ldl: for i = 0 .. n-2, every later entry j subtracts v(i) l(j, i).
// Lane l keeps entries l and l + 64; columns are fetched three steps ahead like the rows above.
// Writes v to vv.  All 64 lanes call it.
struct UnitLower {
    const double *a;
    double *vv;
    int n, lane, l0, l1;
    double v0, v1;
    double c0[4], c1[4];

    __device__ __forceinline__ void fetch(int i, int q)     // column i into set q (past the last step: the last again)
    {
        const int ii = i < n - 2 ? i : (n >= 2 ? n - 2 : 0);
        const double *col = a + LSQW_EO(ii) - ii;             // col[j] = l(j, ii)
        c0[q] = col[l0];
        c1[q] = col[l1];
    }
    // Step i: v(i) sits in lane i % 64 of v0 (kLow: i < 64) or v1; entries above i subtract v(i) l(j, i).
    // The entries up to i are finished, and nothing reads them from v0 / v1 again: they are not protected from the
    // subtraction (a select between the subtraction and the next step's v_readlane costs ~40 cycles, and so does one
    // that files v(i) away in its lane); v(i) goes to vv as it passes through the scalar unit.
    template <bool kLow>
    __device__ __forceinline__ void step(int i, int q)
    {
        const double v = bcast(kLow ? v0 : v1, i & 63);
        vv[i] = v;                                        // every lane stores the same value: no exec mask to change
        if (kLow) v0 = v0 - v * c0[q];
        v1 = v1 - v * c1[q];
    }
    template <bool kLow>
    __device__ __forceinline__ void four(int i)
    {
        fetch(i + 3, 3); if (i + 1 < n) step<kLow>(i, 0);
        fetch(i + 4, 0); if (i + 2 < n) step<kLow>(i + 1, 1);
        fetch(i + 5, 1); if (i + 3 < n) step<kLow>(i + 2, 2);
        fetch(i + 6, 2); if (i + 4 < n) step<kLow>(i + 3, 3);
    }
};

__device__ __forceinline__ void unit_lower_solve(const double *a, const double *z, double *vv, int n)
{
    UnitLower U;
    U.a = a; U.vv = vv; U.n = n;
    U.lane = lane_here();
    U.l0 = U.lane < n ? U.lane : n - 1;
    U.l1 = U.lane + 64 < n ? U.lane + 64 : n - 1;
    U.v0 = z[U.l0];
    U.v1 = z[U.l1];
    U.fetch(0, 0); U.fetch(1, 1); U.fetch(2, 2);
    int i = 0;
    for (; i + 1 < n && i < 64; i += 4) U.four<true>(i);          // i .. i + 3 < 64: v(i) is in v0
    for (; i + 1 < n; i += 4) U.four<false>(i);
    // the last entry is nobody's pivot: it is finished where it is
    if (U.lane == (n - 1) % 64) vv[n - 1] = n - 1 < 64 ? U.v0 : U.v1;
}

#undef LSQW_EO

}  // namespace lsqwave

#endif  // __HIP_DEVICE_COMPILE__
