// Device-side building blocks shared by all cpengine kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CP_BLOCK 256          // threads per workgroup (4 waves)
#define CP_WAVES (CP_BLOCK / 64)
#define CP_KMAX 16            // max images per mf_nn problem (one 16x16 f64 MFMA tile)
#define CP_KBIG 32            // max images of the mf_nn models on the general path (K > 16: 2 x 2 MFMA tiles, cp_mfnn_big.cuh)
#define CP_VAR_KMAX 64        // max images of the variance models (no K x K spectral work: only the LDS hand-over row scales with K)
#define CP_TS 66              // LDS tile row stride (doubles): conflict-free ds_read_b64 in MFMA order
#define CP_NSUM 12            // per-pass partial sums   (slots 0..11)
#define CP_NMAX 8             // per-pass partial maxima (slots 12..19)
#define CP_NSCAL (CP_NSUM + CP_NMAX)
#define CP_PSTRIDE (CP_NSCAL + 2 * CP_KMAX * CP_KMAX)   // doubles per block in the partial buffer: scalars + 2 Grams

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// wave / block reductions (deterministic: fixed shuffle tree, fixed wave order)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// Reduce NS sums and NM maxima over the block; thread 0 writes them to out[0..NS+NM).
// red: LDS scratch of at least NWV * (NS + NM) doubles (NWV = waves of the workgroup).
template <int NS, int NM, int NWV = CP_WAVES>
__device__ __forceinline__ void block_reduce_store(const double (&s)[NS], const double (&mx)[NM],
                                                   double *red, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NS; ++a) {
        double v = wave_sum(s[a]);
        if (lane == 0) red[wave * (NS + NM) + a] = v;
    }
#pragma unroll
    for (int a = 0; a < NM; ++a) {
        double v = wave_max(mx[a]);
        if (lane == 0) red[wave * (NS + NM) + NS + a] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS + NM) {
        const int a = threadIdx.x;
        double v = red[a];
        for (int w = 1; w < NWV; ++w) {
            const double o = red[w * (NS + NM) + a];
            v = (a < NS) ? v + o : fmax(v, o);
        }
        out[a] = v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Gram contraction on the matrix cores.
// tile: wave-private LDS image [16][CP_TS]; tile[img][p] = value of image `img` at the wave's p-th
// pixel (rows >= K and pixels past the end hold zeros).  Accumulates Y^T Y of the 64 x 16 block
// into acc: acc[r] = G[(lane>>4) + 4r][lane&15]  (v_mfma_f64_16x16x4_f64 C/D layout).
__device__ __forceinline__ void gram_tile_accum(const double *tile, double4_t &acc) {
    const int lane = threadIdx.x & 63;
    const double *src = tile + (lane & 15) * CP_TS + (lane >> 4);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const double v = src[4 * t];   // A[img][pix] == B[pix][img]: same register for both operands
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
    }
}

// The same for KM = 16 * NT images (K > 16: NT = 2): tile [KM][CP_TS], NT x NT accumulators,
// acc.a[a * NT + b][r] = G[16 a + (lane>>4) + 4r][16 b + (lane&15)].
template <int NT>
struct GramAcc { double4_t a[NT * NT]; };
template <int NT>
__device__ __forceinline__ void gram_acc_zero(GramAcc<NT> &acc) {
#pragma unroll
    for (int q = 0; q < NT * NT; ++q) acc.a[q] = double4_t{0.0, 0.0, 0.0, 0.0};
}
template <int NT>
__device__ __forceinline__ void gram_tile_accum_n(const double *tile, GramAcc<NT> &acc) {
    const int lane = threadIdx.x & 63;
    const double *src = tile + (lane & 15) * CP_TS + (lane >> 4);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        double v[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) v[a] = src[a * 16 * CP_TS + 4 * t];
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc.a[a * NT + b] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[a], v[b], acc.a[a * NT + b], 0, 0, 0);
    }
}
// fixed-order sum over the waves, (16 NT)^2 entries row-major to out[]; stage: LDS scratch of NWV * (16 NT)^2 doubles
template <int NT, int NWV = CP_WAVES>
__device__ __forceinline__ void gram_block_store_n(const GramAcc<NT> &acc, double *stage, double *out) {
    constexpr int KM = 16 * NT, E = KM * KM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * a + (lane >> 4) + 4 * r, col = 16 * b + (lane & 15);
                stage[wave * E + row * KM + col] = acc.a[a * NT + b][r];
            }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += NWV * 64) {
        double v = stage[e];
        for (int w = 1; w < NWV; ++w) v += stage[w * E + e];
        out[e] = v;
    }
    __syncthreads();
}
// doubles per row of the partial buffer for kernels built for up to KM images: scalars + 2 Gram matrices
template <int KM>
__host__ __device__ constexpr int cp_pstride() { return CP_NSCAL + 2 * KM * KM; }

// Sum the 4 waves' Gram accumulators in fixed order and write the 256 entries to out[].
// stage: LDS scratch of CP_WAVES*256 doubles.
template <int NWV = CP_WAVES>
__device__ __forceinline__ void gram_block_store(const double4_t &acc, double *stage, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = (lane >> 4) + 4 * r, col = lane & 15;
        stage[wave * 256 + row * 16 + col] = acc[r];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int e = threadIdx.x;   // 256 entries
        double v = stage[e];
        for (int w = 1; w < NWV; ++w) v += stage[w * 256 + e];
        out[e] = v;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// cos/sin of atan2(y, x) / 2 without transcendentals (the angle lies in [-pi/2, pi/2]).
// atan2(0, 0) = 0 as in MATLAB (NNE/svd_2x2.m:22,30).
__device__ __forceinline__ void half_angle(double x, double y, double &c, double &s) {
    const double r = sqrt(x * x + y * y);
    if (r == 0.0) { c = 1.0; s = 0.0; return; }
    const double inv2r = 0.5 / r;
    if (x >= 0.0) {
        c = sqrt((r + x) * inv2r);
        s = y * inv2r / c;
    } else {
        s = copysign(sqrt((r - x) * inv2r), y);
        c = y * inv2r / s;
    }
}

// singular values of the 2x2 matrix [a b; c d] given as (a, c, b, d) = (m11, m21, m12, m22)
// NNE/svd_2x2.m:15-20
__device__ __forceinline__ void sv_2x2(double a, double c, double b, double d, double &sig1, double &sig2) {
    const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    const double s1 = aa + bb + cc + dd;
    const double xt = aa + bb - cc - dd;
    const double acbd = a * c + b * d;
    const double s2 = sqrt(xt * xt + 4.0 * (acbd * acbd));
    sig1 = sqrt(fmax((s1 + s2) * 0.5, 0.0));
    sig2 = sqrt(fmax((s1 - s2) * 0.5, 0.0));
}

__device__ __forceinline__ double sgn(double v) { return (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0); }

// U * diag(max(Sigma - thr, 0)) * V' of the 2x2 matrix (a, c, b, d), following NNE/svd_2x2.m:22-47 and
// NNE/pointwise_2x2_nn.m:16-24 (including the sign(0) = 0 columns of V in degenerate cases).
__device__ __forceinline__ void svt_2x2(double a, double c, double b, double d, double thr,
                                        double &r11, double &r21, double &r12, double &r22) {
    const double aa = a * a, bb = b * b, cc = c * c, dd = d * d;
    const double s1 = aa + bb + cc + dd;
    const double xt = aa + bb - cc - dd;
    const double yt = 2.0 * a * c + 2.0 * b * d;
    const double acbd = a * c + b * d;
    const double s2 = sqrt(xt * xt + 4.0 * (acbd * acbd));
    const double sig1 = sqrt(fmax((s1 + s2) * 0.5, 0.0));
    const double sig2 = sqrt(fmax((s1 - s2) * 0.5, 0.0));
    const double t1 = fmax(sig1 - thr, 0.0), t2 = fmax(sig2 - thr, 0.0);
    if (t1 == 0.0 && t2 == 0.0) { r11 = r21 = r12 = r22 = 0.0; return; }
    double ct, st, cp, sp;
    half_angle(xt, yt, ct, st);
    const double xp = aa - bb + cc - dd;
    const double yp = 2.0 * a * b + 2.0 * c * d;
    half_angle(xp, yp, cp, sp);
    // U = [u11 u21 u12 u22] = [ct st -st ct]
    const double s11 = (a * ct + c * st) * cp + (b * ct + d * st) * sp;
    const double s22 = (a * st - c * ct) * sp + (-b * st + d * ct) * cp;
    const double g1 = sgn(s11), g2 = sgn(s22);
    const double v11 = g1 * cp, v21 = g1 * sp, v12 = -g2 * sp, v22 = g2 * cp;
    r11 = t1 * ct * v11 + t2 * (-st) * v12;
    r21 = t1 * st * v11 + t2 * ct * v12;
    r12 = t1 * ct * v21 + t2 * (-st) * v22;
    r22 = t1 * st * v21 + t2 * ct * v22;
}

// ---------------------------------------------------------------------------------------------
// Uniform base (SGPR pair) + 32-bit per-lane byte offset + immediate: the global_load/store "saddr" form. One VGPR per
// distinct offset instead of a 64-bit VGPR address per access, and no per-access 64-bit VALU address arithmetic.
// `ubase` must be wave-uniform; a field is < 4 GiB so the byte offset fits 32 bits.
// S = storage type of a field (double, or float in the fp32 mode); values are always returned / taken as double.
template <typename S>
__device__ __forceinline__ double rnd(double v) { return (double)(S)v; }   // what a store of v keeps
template <typename S>
__device__ __forceinline__ double ldo(const S *__restrict__ ubase, unsigned off, int delta = 0) {
    return (double)*reinterpret_cast<const S *>(reinterpret_cast<const char *>(ubase) + (size_t)off + (ptrdiff_t)delta);
}
template <typename S>
__device__ __forceinline__ void sto(S *__restrict__ ubase, unsigned off, double v) {
    // streaming store: results are not read again by the launch that writes them, so they should not displace the
    // neighbour lines other workgroups are about to re-read from L2 (measured: -2 GB fetched, -1.5..3 % time)
    __builtin_nontemporal_store((S)v, reinterpret_cast<S *>(reinterpret_cast<char *>(ubase) + (size_t)off));
}
// single-use streams (read by exactly one thread of the launch)
template <typename S>
__device__ __forceinline__ double ldo_stream(const S *__restrict__ ubase, unsigned off) {
    return (double)__builtin_nontemporal_load(reinterpret_cast<const S *>(reinterpret_cast<const char *>(ubase) + (size_t)off));
}

// ---------------------------------------------------------------------------------------------
// Buffer addressing (round 2): a 128-bit resource descriptor in SGPRs (uniform base of ONE field group) + a 32-bit per-lane
// byte offset (VGPR) + a uniform byte offset (SGPR) + the instruction's cache bits. What ldo()/sto() above only ask for, the
// buffer instructions guarantee: the compiler turned `uniform base + field*fs + zext(off)` of the fused NN-TV kernel into
// ~25 loop-invariant 64-bit per-lane offsets (50 VGPRs) plus a 64-bit VALU add per access, and -- at the 256-VGPR limit of
// two waves per SIMD -- issued the 38 loads of an image in five dependent groups. With descriptors every position is ONE
// VGPR, and all loads of an image are in flight before the first wait.
// num_records = 2^32 - 1: the range check (voffset + imm against num_records; soffset is not checked) never fires.
typedef __amdgpu_buffer_rsrc_t cp_rsrc;
typedef unsigned int cp_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cp_rsrc mk_rsrc(const void *uniform_base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(uniform_base), 0, -1, 0x00020000);
}
template <typename S>
__device__ __forceinline__ double bld(cp_rsrc r, unsigned voff, unsigned soff = 0u) {
    if constexpr (sizeof(S) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    else return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// the stored value as it is (fp32 mode: the conversion to double is left to the consumer, so that a group of loads has
// distinct one-register destinations and none of them waits for the one before)
template <typename S>
__device__ __forceinline__ S bld_raw(cp_rsrc r, unsigned voff, unsigned soff = 0u) {
    if constexpr (sizeof(S) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <typename S>
__device__ __forceinline__ S bld_raw_stream(cp_rsrc r, unsigned voff, unsigned soff = 0u) {
    if constexpr (sizeof(S) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
}
template <typename S>
__device__ __forceinline__ double bld_stream(cp_rsrc r, unsigned voff, unsigned soff = 0u) {   // nt: read once by the launch
    if constexpr (sizeof(S) == 8) return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2));
    else return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
}
template <typename S>
__device__ __forceinline__ void bst(cp_rsrc r, unsigned voff, unsigned soff, double v) {        // nt streaming store, cf. sto()
    if constexpr (sizeof(S) == 8) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cp_u2, v), r, voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), r, voff, soff, 2);
}

// two consecutive elements per lane (16 bytes in fp64 storage, 8 in fp32): k_curv_y2_r2
typedef unsigned int cp_u4 __attribute__((ext_vector_type(4)));
template <typename S>
struct cp_pair { S a, b; };
// The loaded bits are reinterpreted as the pair in one bit_cast. (Taking the two halves of the b64 builtin's result as vector
// elements -- cp_u2 v = load; v.x, v.y -- compiled to ONE buffer_load_dword with both halves equal on this toolchain.)
template <typename S, int CACHE>
__device__ __forceinline__ cp_pair<S> bld2_raw(cp_rsrc r, unsigned voff) {
    if constexpr (sizeof(S) == 8) return __builtin_bit_cast(cp_pair<S>, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0u, CACHE));
    else return __builtin_bit_cast(cp_pair<S>, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0u, CACHE));
}
template <typename S>
__device__ __forceinline__ void bst2(cp_rsrc r, unsigned voff, double a, double b) {            // nt streaming store
    if constexpr (sizeof(S) == 8) {
        const cp_u2 lo = __builtin_bit_cast(cp_u2, a), hi = __builtin_bit_cast(cp_u2, b);
        __builtin_amdgcn_raw_buffer_store_b128(cp_u4{lo.x, lo.y, hi.x, hi.y}, r, voff, 0u, 2);
    } else {
        __builtin_amdgcn_raw_buffer_store_b64(cp_u2{__builtin_bit_cast(unsigned, (float)a), __builtin_bit_cast(unsigned, (float)b)}, r, voff, 0u, 2);
    }
}

// sqrt(x) and 1/sqrt(x) from one v_rsq_f64 and the Goldschmidt / residual steps of the compiler's own f64 sqrt
// expansion, without its range scaling: arguments here are sums of squares of image-scale quantities. Below 1e-280
// (in practice: exactly 0) both results are 0.
__device__ __forceinline__ void sqrt_rsqrt(double x, double &sq, double &inv) {
    const bool ok = x >= 1.0e-280;
    const double xs = ok ? x : 1.0;
    const double y0 = __builtin_amdgcn_rsq(xs);
    double g = xs * y0, h = 0.5 * y0;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, xs);
    g = fma(d, h, g);
    d = fma(-g, g, xs);
    g = fma(d, h, g);                       // sqrt(x), correctly rounded up to the last step's rounding
    double iv = h + h;
    const double e = fma(-g, iv, 1.0);
    iv = fma(iv, e, iv);                    // 1/sqrt(x) to ~1 ulp
    sq = ok ? g : 0.0;
    inv = ok ? iv : 0.0;
}

// 2x2 matrices M = (m11, m21, m12, m22) split into a scaled rotation [E -H; H E]/2 and a scaled reflection [F G; G -F]/2
// with E = m11+m22, H = m21-m12, F = m11-m22, G = m21+m12. With Q = |(E,H)|, R = |(F,G)|: the singular values are
// (Q+R)/2 and |Q-R|/2, the singular vectors depend only on the two directions. Hence
//   nuclear norm            = max(Q, R)
//   U diag(f1, f2) V'       = p/Q [E -H; H E] + q/R [F G; G -F],  p = (f1 + d f2)/2, q = (f1 - d f2)/2, d = sign(Q - R)
// for any f that is a function of the singular values (unique when Q or R vanish since then f1 = f2 matters only through
// the surviving part). Same values as NNE/svd_2x2.m + pointwise_2x2_nn.m up to rounding, from 2 rsqrt instead of 10 sqrt
// and 7 divisions.
__device__ __forceinline__ double nuc_2x2(double m11, double m21, double m12, double m22) {
    const double E = m11 + m22, H = m21 - m12, F = m11 - m22, G = m21 + m12;
    double sq, inv;
    sqrt_rsqrt(fmax(fma(E, E, H * H), fma(F, F, G * G)), sq, inv);
    return sq;
}

// y - sigma * prox_{(mu/sigma)||.||_*}(y / sigma) = projection of y onto {largest singular value <= mu}
// (NNE/pointwise_2x2_nn.m:46-48 via Moreau, positively homogeneous in sigma). s1 = largest singular value of the result.
__device__ __forceinline__ void proj_spectral_2x2(double m11, double m21, double m12, double m22, double mu,
                                                  double &r11, double &r21, double &r12, double &r22, double &s1) {
    const double E = m11 + m22, H = m21 - m12, F = m11 - m22, G = m21 + m12;
    double Q, iQ, R, iR;
    sqrt_rsqrt(fma(E, E, H * H), Q, iQ);
    sqrt_rsqrt(fma(F, F, G * G), R, iR);
    const double sv1 = 0.5 * (Q + R), sv2 = 0.5 * fabs(Q - R);
    const bool inside = sv1 <= mu;
    const double f2 = fmin(sv2, mu);
    const double df2 = (Q >= R) ? f2 : -f2;
    const double cq = 0.5 * (mu + df2) * iQ, cr = 0.5 * (mu - df2) * iR;
    const double qe = cq * E, qh = cq * H;
    r11 = inside ? m11 : fma(cr, F, qe);
    r22 = inside ? m22 : fma(-cr, F, qe);
    r21 = inside ? m21 : fma(cr, G, qh);
    r12 = inside ? m12 : fma(cr, G, -qh);
    s1 = inside ? sv1 : mu;
}

// row sum of the 5-point Laplacian at pixel (i, jg): 0 in the interior and for neumann rows
// (NNE/discrete_laplacian.m:25-27: diagonal -1/h^2 at the ends), -1/h^2 per missing neighbour for dirichlet rows (:28)
__device__ __forceinline__ double lap_rowsum(bool neumann, bool first_i, bool last_i, bool first_j, bool last_j, double ih1,
                                             double ih2) {
    if (neumann) return 0.0;
    return -((first_i ? 1.0 : 0.0) + (last_i ? 1.0 : 0.0)) * ih1 - ((first_j ? 1.0 : 0.0) + (last_j ? 1.0 : 0.0)) * ih2;
}

// ---------------------------------------------------------------------------------------------
// forward difference along one axis with the reference's boundary rows
// (NNE/finite_difference_operator.m:29-36): interior (u_next - u)/h; last row neumann 0,
// linear u - u_prev (unscaled).
__device__ __forceinline__ double fwd_diff(double u, double u_next, double u_prev, bool last, bool linear,
                                           double inv_h) {
    return last ? (linear ? (u - u_prev) : 0.0) : (u_next - u) * inv_h;
}

// adjoint of the above at position c of an axis with `len` entries: (D^T w)(c).
// w_c = w(c), w_m = w(c-1), w_last = w(len-1).
__device__ __forceinline__ double fwd_diff_adj(double w_c, double w_m, double w_last, int c, int len, bool linear,
                                               double inv_h) {
    double r = 0.0;
    if (c >= 1) r += w_m * inv_h;
    if (c <= len - 2) r -= w_c * inv_h;
    if (linear) {
        if (c == len - 2) r -= w_last;
        if (c == len - 1) r += w_last;
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// bilinear interpolation with zero padding and its analytic derivatives at the world point (px, py)
// (IRE/bilinear_interpolation.m:29-105). Coordinate arithmetic uses explicitly rounded (non-fused)
// operations so that floor() -- the index logic -- is bit-identical to the CPU restatement.
__device__ __forceinline__ void bilinear_tap(const double *__restrict__ img, int mi, int ni, double px, double py,
                                             double inv_h1, double inv_h2, double &val, double &dx, double &dy) {
    const double q1 = __dadd_rn(__dmul_rn(inv_h1, px), 0.5), q2 = __dadd_rn(__dmul_rn(inv_h2, py), 0.5);   // :29-35
    const double f1 = floor(q1), f2 = floor(q2);                                                           // :42
    const double c1 = q1 - f1, c2 = q2 - f2;                                                               // :64
    val = 0.0; dx = 0.0; dy = 0.0;
    // clamp before the int conversion: far out-of-domain samples have no valid tap anyway
    const double lim = 1.0e9;
    const long long s1 = (long long)fmin(fmax(f1, -lim), lim), s2 = (long long)fmin(fmax(f2, -lim), lim);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int di = t & 1, dj = t >> 1;
        const long long si = s1 + di, sj = s2 + dj;                     // 1-based subscripts
        if (si >= 1 && si <= mi && sj >= 1 && sj <= ni) {               // zero padding (:48-55)
            const double v = img[(size_t)(sj - 1) * mi + (si - 1)];
            const double w1 = di ? c1 : (1.0 - c1), w2 = dj ? c2 : (1.0 - c2);
            val += w1 * w2 * v;
            dx += inv_h1 * ((di ? 1.0 : -1.0) * w2) * v;                // :71-72,81-82,91-92,101-102
            dy += inv_h2 * ((dj ? 1.0 : -1.0) * w1) * v;                // :74-75,84-85,94-95,104-105
        }
    }
}

