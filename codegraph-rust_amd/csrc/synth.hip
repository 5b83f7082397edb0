// SURVEY.md §8(d) synthetic inputs: a COUNTER-BASED generator keyed (seed, row, col) -> N(0, 1) f32 -> L2-normalised in f32, so
// that the device and the CPU checker of tests / bench (cgo_synth_rows, a separate restatement of this contract) produce
// bit-identical corpora of any size without shipping them, in any chunking and on any number of ranks. The reference has no such
// generator (its benches use `rand`); this is measurement plumbing behind the C ABI, not part of the replaced path.
//
// The contract (every operation is one IEEE f32 rounding; the library is built -ffp-contract=off with correctly rounded / and sqrt):
//   block b = col / 4 of row r:  (x0, x1, x2, x3) = Philox4x32-10(counter = (b, r_lo, r_hi, 0), key = (seed_lo, seed_hi))
//   two Box-Muller pairs: (x0, x1) -> cols 4b, 4b+1;  (x2, x3) -> cols 4b+2, 4b+3
//       u = ((xa >> 9) + 0.5) * 2^-23  in (0, 1), exact;   v = (xb >> 8) * 2^-24  in [0, 1), exact
//       rad = sqrt(-2 * ln(u));  z_even = rad * cos(2 pi v);  z_odd = rad * sin(2 pi v)
//     ln / sin / cos are the POLYNOMIALS below (own range reduction, Horner in f32), not libm: two libm's differ in the last bit
//   norm^2: lane l (0..63) sums the squares of its blocks l, l+64, ... in column order (acc = acc + z * z), then the 64 partial
//     sums meet in a xor butterfly (offsets 32, 16, 8, 4, 2, 1: acc = acc + partner); every element is DIVIDED by sqrt(norm^2).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/cgvec.h"

extern "C" int cgv_set_error_(int code, const char* msg);  // abi.hip

namespace {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// ln(u), u in (0, 1) a normal f32: u = m * 2^e with m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(t), t = (m - 1) / (m + 1), |t| <= 0.1716
__device__ __forceinline__ float synth_ln(float u) {
    uint32_t b = __float_as_uint(u);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);   // [1, 2)
    if (m > 1.41421356f) {
        m = m * 0.5f;
        e += 1;
    }
    const float t = (m - 1.0f) / (m + 1.0f);
    const float t2 = t * t;
    float p = 0.0909090936f;          // 1/11
    p = p * t2 + 0.111111112f;        // 1/9
    p = p * t2 + 0.142857149f;        // 1/7
    p = p * t2 + 0.2f;
    p = p * t2 + 0.333333343f;
    p = p * t2 + 1.0f;
    return (float)e * 0.693147182f + (2.0f * t) * p;
}

// (cos, sin)(2 pi v), v in [0, 1): quadrant q = floor(4 v), angle a = (4 v - q) * pi / 2 in [0, pi/2), Taylor polynomials in a^2
__device__ __forceinline__ void synth_sincos2pi(float v, float& c, float& s) {
    const float a4 = v * 4.0f;
    const int q = (int)a4;
    const float a = (a4 - (float)q) * 1.57079637f;
    const float a2 = a * a;
    float ps = -2.50521084e-08f;      // -1/11!
    ps = ps * a2 + 2.75573188e-06f;   //  1/9!
    ps = ps * a2 - 1.98412701e-04f;   // -1/7!
    ps = ps * a2 + 8.33333377e-03f;   //  1/5!
    ps = ps * a2 - 0.166666672f;      // -1/3!
    ps = ps * a2 + 1.0f;
    const float sn = a * ps;
    float pc = 2.08767570e-09f;       //  1/12!
    pc = pc * a2 - 2.75573188e-07f;   // -1/10!
    pc = pc * a2 + 2.48015876e-05f;   //  1/8!
    pc = pc * a2 - 1.38888892e-03f;   // -1/6!
    pc = pc * a2 + 4.16666679e-02f;   //  1/4!
    pc = pc * a2 - 0.5f;
    pc = pc * a2 + 1.0f;
    const float cs = pc;
    switch (q & 3) {
        case 0: c = cs; s = sn; break;
        case 1: c = -sn; s = cs; break;
        case 2: c = -cs; s = -sn; break;
        default: c = sn; s = -cs; break;
    }
}

__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& z0, float& z1) {
    const float u = ((float)(xa >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float v = (float)(xb >> 8) * 5.9604644775390625e-08f;
    const float rad = sqrtf(-2.0f * synth_ln(u));
    float c, s;
    synth_sincos2pi(v, c, s);
    z0 = rad * c;
    z1 = rad * s;
}

// one wave per row; `normalise` = 0 leaves the N(0, 1) values (tests of the generator's distribution)
__global__ __launch_bounds__(256) void synth_rows_kernel(float* __restrict__ out, uint64_t seed, uint64_t row0, uint64_t nrows,
                                                         uint32_t dim, int normalise) {
    const int lane = threadIdx.x & 63;
    const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nrows) return;
    const uint64_t r = row0 + i;
    float* o = out + i * dim;
    const uint32_t nblk = (dim + 3) / 4;
    float acc = 0.0f;
    for (uint32_t b = lane; b < nblk; b += 64) {
        uint32_t x[4];
        philox4x32_10(b, (uint32_t)r, (uint32_t)(r >> 32), 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        float z[4];
        box_muller(x[0], x[1], z[0], z[1]);
        box_muller(x[2], x[3], z[2], z[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < dim) {
                o[4 * b + j] = z[j];
                acc = acc + z[j] * z[j];
            }
    }
    if (!normalise) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off, 64);
    const float nrm = sqrtf(acc);
    __builtin_amdgcn_s_waitcnt(0);   // (a lane reads back only what it wrote itself)
    for (uint32_t b = lane; b < nblk; b += 64)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * b + j < dim) o[4 * b + j] = o[4 * b + j] / nrm;
}

}  // namespace

extern "C" int cgv_synth_rows_f32_dev(int device_id, uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, int normalise,
                                      float* out_dev, void* stream) {
    if (nrows == 0 || dim == 0) return CGV_OK;
    if (!out_dev) return cgv_set_error_(CGV_ERR_INVALID_ARG, "out is NULL");
    if (cgv_device_count() == 0) return cgv_set_error_(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(synth_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, out_dev, seed, row0,
                           nrows, dim, normalise);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return cgv_set_error_(CGV_ERR_HIP, (std::string("cgv_synth_rows_f32_dev: ") + hipGetErrorString(e)).c_str());
    return CGV_OK;
}
