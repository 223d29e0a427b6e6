// loglik_math.h -- the Gaussian-approximated Poisson log-probability of one sinogram sample (a8,
// ctvae/helper_functions.py:360-368), shared by the elementwise kernel (loglik.hip) and the planned forward's fused
// epilogue (rotate_plan.hip) so that both evaluate the same fp32 expression.
#pragma once
#include <hip/hip_runtime.h>

namespace ctpvae {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// tfd.Normal(loc = proj * mask, scale = eps + sqrt(loc / pnm + eps)).log_prob(x), in tfp's form
//   -0.5 * (x / scale - loc / scale)^2 - (0.5 * log(2 pi) + log(scale))
// -- with its three IEEE divisions as they stand: where x and loc nearly cancel, the value IS the rounding of those two large
// quotients (x / scale ~ 1e2 .. 1e4, so ~1e-5 absolute), and only the same operations on the same bits reproduce it (taken as
// products with 1 / scale the golden log-probabilities moved by 6e-6 of their largest value, 2e-5 in single samples: reverted).
//
// The divisions are IEEE quotients WITHOUT hipcc's twelve-instruction sequence per quotient: that sequence is (v_div_scale x 2,
// v_rcp, two Newton steps on the reciprocal, quotient, two residual corrections -- the last through v_div_fmas --, v_div_fixup), and
// for operands it does not have to rescale (finite, the quotient far from the denormals: every quotient here -- scale >= eps > 0,
// pnm > 0) scale / fmas / fixup are the identity.  div_by() is the rest of it, verbatim, with the refined reciprocal of the
// denominator taken ONCE for the quotients that share it (x / scale and loc / scale; loc / pnm of every sample of a kernel): the
// same bits (tools/ab_loglik_div.py: 2^26 samples against the compiler's sequence, 0 differ), 5 instead of 12 instructions per quotient.
struct Recip {
    float d, r;   // a denominator and its reciprocal, v_rcp_f32 + one Newton step (the IEEE sequence's r1)
};
__device__ __forceinline__ Recip recip_refined(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    return Recip{d, r};
}
__device__ __forceinline__ float div_by(float a, const Recip &k)
{
    float q = a * k.r;
    float e = __builtin_fmaf(-k.d, q, a);
    q = __builtin_fmaf(e, k.r, q);
    e = __builtin_fmaf(-k.d, q, a);
    return __builtin_fmaf(e, k.r, q);
}
struct GaussPoisson {
    float loc, root, scale;
    Recip rscale;
};
__device__ __forceinline__ GaussPoisson gaussian_poisson_terms(float proj, float m, float pnm, float eps)
{
    GaussPoisson t;
    t.loc = proj * m;
    t.root = sqrtf(div_by(t.loc, recip_refined(pnm)) + eps);
    t.scale = eps + t.root;
    t.rscale = recip_refined(t.scale);
    return t;
}
__device__ __forceinline__ float gaussian_poisson_logp(const GaussPoisson &t, float x)
{
    const float z = div_by(x, t.rscale) - div_by(t.loc, t.rscale);
    return -0.5f * (z * z) - (kHalfLog2Pi + logf(t.scale));
}
__device__ __forceinline__ float gaussian_poisson_logp(float proj, float m, float x, float pnm, float eps)
{
    return gaussian_poisson_logp(gaussian_poisson_terms(proj, m, pnm, eps), x);
}

// d logp / d proj (the mask factor included) and d logp / d pnm of the same sample: what the backward multiplies the
// upstream gradient by.  One expression for loglik_bwd_kernel and for the projector epilogue that stores dlp.  Nothing pins
// its bits (the tests hold it to float64 autograd at 1e-4), so its quotients are products with TWO reciprocals, 1 / scale (the log-probability's refined one) and
// 1 / root (v_rcp_f32), and with the caller's 1 / pnm (once per kernel) -- written with a division per quotient (round 2) the derivative
// was six divisions and a second root, ~110 of the ~180 vector instructions a sample cost.  (Measured on one box: config 5's
// forward + likelihood + sums 122.4 -> 122.2 us, the training call 9.9 -> 9.85 us -- the epilogues' arithmetic hides under
// their memory traffic; kept because it is less code, not because it is faster.)
__device__ __forceinline__ float gaussian_poisson_dlogp(const GaussPoisson &t, float m, float x, float inv_pnm, float &dpnm)
{
    const float rs = t.rscale.r;                         // (the refined reciprocal: within an ulp of 1 / scale)
    const float z = (x - t.loc) * rs;
    const float dscale = (z * z - 1.0f) * rs;            // d logp / d scale
    const float dscale_du = 0.5f * __builtin_amdgcn_rcpf(t.root);   // d scale / d (loc/pnm + eps)
    dpnm = dscale * dscale_du * (-t.loc * (inv_pnm * inv_pnm));
    return (z * rs + dscale * dscale_du * inv_pnm) * m;
}
__device__ __forceinline__ float gaussian_poisson_dlogp(float proj, float m, float x, float pnm, float eps, float &dpnm)
{
    return gaussian_poisson_dlogp(gaussian_poisson_terms(proj, m, pnm, eps), m, x, 1.0f / pnm, dpnm);
}

// what a projector kernel needs to write log-probabilities next to its ray-sums (lp == nullptr: no epilogue)
struct LogLikEpilogue {
    const float *mask, *meas, *pnm;   // [S][A], [S][A][PW], one value
    float eps;
    float *lp;                        // [S][A][PW]
    float *dlp;                       // [S][A][PW] d lp / d ray-sum for the backward, or nullptr
    // angle-subset launches only: mask / meas are the DENSE [S][A_plan] / [S][A_plan][PW] arrays, indexed by the plan
    // angle (the caller's gather mask[:, angles_i], proj_sample[:, angles_i] -- ctvae/helper_functions.py:356-357 --
    // folded into the load); 0: they are compact like the outputs
    int dense = 0;
    // per-object sums ("reduction to per-object log-lik", ctvae/helper_functions.py:305-312): when `part` is set
    // the kernel writes ONE partial sum per 64-lane task into part[(s * A_out + k) * tasks_per_row + task] (lanes added by
    // the xor butterfly 32, 16, 8, 4, 2, 1; lanes without a bin add +0.0f) and lp / the ray-sum store become optional; a
    // second, tiny launch adds a slice's partials in the fixed order of object_sum_of_parts below.
    float *part = nullptr;
    // round 4 (knob FOLD_SUMS = 1; measured slower than the second launch, so not the default): the ordered sum of a slice's
    // partials happens INSIDE the launch -- the workgroup that finishes a slice (pair / group)
    // last, told by an arrival counter, reads the partials back and adds them in object_sum_of_parts' order (fixed by the
    // reader, not by arrival: the same bits as the separate loglik_sum_partials_kernel launch of round 3).  sum [S]; arrive: one
    // zero-initialised counter per unit, left zero again.  Partials cross workgroups (and XCDs) inside one launch, so they are
    // stored and re-read at agent scope (sc1) behind the arrival add -- MI355X_MICROARCH.md's counter hand-off, no L2 fence.
    float *sum = nullptr;
    unsigned *arrive = nullptr;
    __device__ __forceinline__ void store_part(size_t i, float v) const
    {
        if (sum != nullptr) __hip_atomic_store(part + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else part[i] = v;   // read by the next launch: a plain store
    }

    // o: offset of the ray-sum in the outputs; om / sa: offsets of its measured sample and its mask entry
    __device__ __forceinline__ void write(size_t o, size_t om, size_t sa, float raysum) const
    {
        const float pnm_v = *pnm;
        write_loaded(o, mask[sa], meas[om], pnm_v, 1.0f / pnm_v, raysum);
    }
    // ... with the operands already in registers (pnm_v = *pnm and its reciprocal, taken once per kernel): a kernel requests them
    // BEFORE its long phase (the walk, the sum over tiles) so that their round trip to memory is not paid after it -- the
    // stores below keep the compiler from moving the loads up by itself (nothing tells it that the buffers are distinct)
    __device__ __forceinline__ void write_loaded(size_t o, float m, float x, float pnm_v, float inv_pnm, float raysum) const
    {
        (void)eval_loaded(o, m, x, pnm_v, inv_pnm, raysum);
    }
    // the same, returning the log-probability; lp (and dlp) are stored only where a buffer was given
    __device__ __forceinline__ float eval(size_t o, size_t om, size_t sa, float raysum) const
    {
        const float pnm_v = *pnm;
        return eval_loaded(o, mask[sa], meas[om], pnm_v, 1.0f / pnm_v, raysum);
    }
    __device__ __forceinline__ float eval_loaded(size_t o, float m, float x, float pnm_v, float inv_pnm, float raysum) const
    {
        const GaussPoisson t = gaussian_poisson_terms(raysum, m, pnm_v, eps);
        const float v = gaussian_poisson_logp(t, x);
        if (lp) lp[o] = v;
        if (dlp) {
            float unused;
            dlp[o] = gaussian_poisson_dlogp(t, m, x, inv_pnm, unused);
        }
        return v;
    }
};

// The fixed order of a 64-lane task's partial sum: xor butterfly, every lane ends with the same value (a + b == b + a bit
// for bit, so both partners of a step compute the same sum).  oracle/radon_oracle.py loglik_object_sums restates it.
// (round 5: the partners' values arrive through the vector unit -- v_permlane32_swap / v_permlane16_swap for the distances 32 and 16,
// DPP row rotations and quad permutations below -- instead of six dependent ds_bpermute round trips through the LDS queue; the same
// six sums a + b, the same bits: a copy pair swapped half against half holds both partners of every lane, and a + b == b + a.)
template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ float wave_sum_dpp(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
    {   // xor 32: lanes 32-63 of the first copy change places with lanes 0-31 of the second: a = {lo, lo}, b = {hi, hi}
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // (two wait states behind the write of v)
        v = a + b;
    }
    {   // xor 16: odd rows of the first copy change places with even rows of the second: a = rows {0, 0, 2, 2}, b = rows {1, 1, 3, 3}
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
        v = a + b;
    }
    v = v + wave_sum_dpp<0x128>(v, v);              // row_ror:8 = xor 8 inside a row of 16
    {
        float p = wave_sum_dpp<0x104, 0x5>(v, v);   // row_shl:4: lanes with bit 2 clear (banks 0, 2) take lane + 4
        p = wave_sum_dpp<0x114, 0xa>(p, v);         // row_shr:4: the others take lane - 4
        v = v + p;
    }
    v = v + wave_sum_dpp<0x4e>(v, v);               // quad_perm [2, 3, 0, 1]
    v = v + wave_sum_dpp<0xb1>(v, v);               // quad_perm [1, 0, 3, 2]
    return v;
}

// The fixed order of the per-object sum over task sums part[a][k] (A angles x tpr tasks per angle):
//   S_a   = ((0 + part[a][0]) + part[a][1]) + ...            one angle's tasks, ascending
//   total = ((0 + B_0) + B_1) + ...,  B_g = wave_sum over lanes l of S_(64 g + l)  (angles past A add +0.0f)
// -- every level is either a short sequential sum or the xor butterfly, so a wave computes it in a few hundred cycles whatever
// A is (a plain ascending sum over all A * tpr task sums took 40 us at 90 angles x 12 tasks: 1080 dependent adds).
// (AGENT: the partials were stored by other workgroups of this launch -- agent-scope loads, see LogLikEpilogue::store_part)
template <bool AGENT = false>
__device__ __forceinline__ float object_sum_of_parts(const float *__restrict__ part, int A, int tpr, int lane)
{
    auto ld = [](const float *q) -> float {
        if constexpr (AGENT) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *q;
    };
    float total = 0.0f;
    if (tpr <= 16) {
        // round 4: ALL of an angle's task sums -- and those of the next group of 64 angles -- are requested before the first add
        // (512 x 512 at 90 angles: 12 tasks per angle in two groups were six dependent batches of four loads, 4.7 us per launch)
        for (int a0 = 0; a0 < A; a0 += 128) {
            const int a = a0 + lane, b = a0 + 64 + lane;
            float v0[16], v1[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                v0[k] = (a < A && k < tpr) ? ld(part + (size_t)a * tpr + k) : 0.0f;
                v1[k] = (b < A && k < tpr) ? ld(part + (size_t)b * tpr + k) : 0.0f;
            }
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < tpr) {   // (wave-uniform) ascending, one at a time
                    s0 += v0[k];
                    s1 += v1[k];
                }
            total += wave_sum(s0);
            if (a0 + 64 < A) total += wave_sum(s1);
        }
        return total;
    }
    for (int a0 = 0; a0 < A; a0 += 64) {
        const int a = a0 + lane;
        float sa = 0.0f;
        if (a < A)
            for (int k = 0; k < tpr; k += 4) {   // four loads in flight, added in order (one at a time: tpr round trips)
                const float *p = part + (size_t)a * tpr + k;
                const int n = tpr - k;
                const float v0 = ld(p), v1 = ld(p + (n > 1 ? 1 : 0)), v2 = ld(p + (n > 2 ? 2 : 0)), v3 = ld(p + (n > 3 ? 3 : 0));
                sa += v0;
                if (n > 1) sa += v1;
                if (n > 2) sa += v2;
                if (n > 3) sa += v3;
            }
        total += wave_sum(sa);
    }
    return total;
}
// The arrival step of the in-launch sum: every wave has waited for its own partial stores, the workgroup meets, one lane adds to
// the unit's counter; true (for the whole workgroup) if this workgroup's add was the last of `expected` -- it may then read every
// partial of the unit at agent scope.  The counter is reset for the next launch.  `flag`: an LDS word of the workgroup.
__device__ __forceinline__ bool arrived_last(unsigned *counter, unsigned expected, volatile int *flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // Round 5: the hand-off is ordered by the memory model, not by what the hardware happens to do -- the arrival
        // add RELEASES this workgroup's partial stores at agent scope (they are behind the barrier above), and the workgroup
        // whose add came last ACQUIRES before it re-reads the others' partials.  (The partials are also stored and re-read sc1,
        // see store_part / object_sum_of_parts<true>: MI355X_MICROARCH.md's counter hand-off.)  This path is off by default.
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + 1u == expected;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *flag = last ? 1 : 0;
    }
    __syncthreads();
    return *flag != 0;
}

// lp_sum[s] of the partial sums a fused epilogue wrote (LogLikEpilogue::part): one wave per slice
[[maybe_unused]] static __global__ __launch_bounds__(64) void loglik_sum_partials_kernel(const float *__restrict__ part, int S, int A,
                                                                                       int tpr, float *__restrict__ out)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const float total = object_sum_of_parts(part + (size_t)s * A * tpr, A, tpr, lane);
    if (lane == 0) out[s] = total;
}

}  // namespace ctpvae
