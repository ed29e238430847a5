// F(2,3) minimal-filtering pseudo-tap table shared by conv1d_f32w.hip and respair_f32w.hip (and, as an enumeration, by pack.py:pack_conv_mf).
#pragma once
// pseudo-tap j of a k-tap filter: accumulator m<acc>, operand x[oa] - x[ob] (op 0), x[oa] + x[ob] (op 1) or x[oa] (op 2); offsets in units of
// the dilation, relative to the even output's first input.  The host's weight order (pack.py:pack_conv_mf) follows the same enumeration.
struct MfTap { int acc, op, oa, ob; };
__host__ __device__ constexpr MfTap mf_tap(int K, int j) {
    const int ng = K / 3, o = 3 * (j / 4);
    if (j < 4 * ng) {
        switch (j & 3) {
            case 0: return MfTap{0, 0, o, o + 2};
            case 1: return MfTap{1, 1, o + 1, o + 2};
            case 2: return MfTap{2, 0, o + 2, o + 1};
            default: return MfTap{3, 0, o + 1, o + 3};
        }
    }
    const int r = j - 4 * ng, ob = 3 * ng;
    if (K % 3 == 1) return r == 0 ? MfTap{0, 2, ob, ob} : MfTap{3, 2, ob + 1, ob + 1};
    return r == 0 ? MfTap{0, 0, ob, ob + 1} : (r == 1 ? MfTap{1, 2, ob + 1, ob + 1} : MfTap{3, 0, ob + 2, ob + 1});
}
__host__ __device__ constexpr int mf_ntaps(int K) { return 4 * (K / 3) + (K % 3 == 1 ? 2 : (K % 3 == 2 ? 3 : 0)); }

