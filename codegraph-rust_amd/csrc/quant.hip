// quant.hip — the reference's ScalarQuantizer / ProductQuantizer on the device (include/cgvec_quant.h).
// Every kernel keeps the reference's f32 operation order where the order matters (sequential distance
// sums, first-minimum-wins assignment, centroid sums in vector order), so the results are bit-identical
// to crates/codegraph-vector/src/persistent.rs:116-477; min / max reductions are order-free.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/cgvec.h"
#include "../../include/cgvec_quant.h"

extern "C" int cgv_set_error_(int code, const char* msg);

namespace {

int fail(int code, const std::string& m) { return cgv_set_error_(code, m.c_str()); }

#define QCHK(expr)                                                                              \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(_e == hipErrorOutOfMemory ? CGV_ERR_OOM : CGV_ERR_HIP,                  \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                     \
    } while (0)

struct DevMem {  // scoped device allocation
    void* p = nullptr;
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <typename T>
    T* as() const {
        return (T*)p;
    }
};

// Rust f32::min / f32::max: the non-NaN operand wins
__device__ inline float rmin(float a, float b) { return (b != b) ? a : ((a != a) ? b : (b < a ? b : a)); }
__device__ inline float rmax(float a, float b) { return (b != b) ? a : ((a != a) ? b : (b > a ? b : a)); }

// per-dimension min / max over n rows: one thread per dimension walks the rows (coalesced across threads)
__global__ void col_minmax_kernel(const float* __restrict__ rows, uint64_t n, uint32_t dim, float* __restrict__ mn,
                                  float* __restrict__ mx) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    float lo = INFINITY, hi = -INFINITY;
    for (uint64_t r = 0; r < n; ++r) {
        const float v = rows[r * dim + d];
        lo = rmin(lo, v);
        hi = rmax(hi, v);
    }
    mn[d] = lo;
    mx[d] = hi;
}

__global__ void sq_encode_kernel(const float* __restrict__ rows, uint64_t n, uint32_t dim, const float* __restrict__ scales,
                                 const float* __restrict__ biases, uint32_t nbits, uint8_t* __restrict__ out) {
    const uint64_t total = n * dim;
    const float max_val = (float)((1u << nbits) - 1u);
    const uint32_t bpv = nbits == 8 ? 1u : (nbits == 16 ? 2u : 4u);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(i % dim);
        const float normalized = (rows[i] - biases[d]) * scales[d];
        float c = fmaxf(normalized, 0.0f);  // f32::max: NaN -> 0.0
        c = fminf(c, max_val);
        const uint32_t q = (uint32_t)c;     // `as u32`: truncation, already in range
        uint8_t* o = out + i * bpv;
        o[0] = (uint8_t)q;
        if (bpv >= 2) o[1] = (uint8_t)(q >> 8);
        if (bpv == 4) {
            o[2] = (uint8_t)(q >> 16);
            o[3] = (uint8_t)(q >> 24);
        }
    }
}

__global__ void sq_decode_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t dim, const float* __restrict__ scales,
                                 const float* __restrict__ biases, uint32_t nbits, float* __restrict__ out) {
    const uint64_t total = n * dim;
    const uint32_t bpv = nbits == 8 ? 1u : (nbits == 16 ? 2u : 4u);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(i % dim);
        const uint8_t* c = codes + i * bpv;
        uint32_t q = c[0];
        if (bpv >= 2) q |= (uint32_t)c[1] << 8;
        if (bpv == 4) q |= ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
        out[i] = (float)q / scales[d] + biases[d];
    }
}

// euclidean_distance (persistent.rs:320-328): sequential sum of (x - y)^2, then sqrt
__device__ inline float pq_dist(const float* a, const float* b, uint32_t dsub) {
    float s = 0.0f;
    for (uint32_t i = 0; i < dsub; ++i) {
        const float d = a[i] - b[i];
        s = s + d * d;
    }
    return sqrtf(s);
}

// nearest centroid of sub-vector `sub` of every row: first strict minimum wins (:196-212, :268-277).
// rows [n][dim]; cent [ksub][dsub] for this sub-quantiser; assign [n] (u32) or codes [n][m] (u8).
__global__ void pq_assign_kernel(const float* __restrict__ rows, uint64_t n, uint32_t dim, uint32_t sub, uint32_t dsub,
                                 const float* __restrict__ cent, uint32_t ksub, uint32_t* __restrict__ assign,
                                 uint8_t* __restrict__ codes, uint32_t m, uint32_t* __restrict__ any_nonzero) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* v = rows + r * dim + (uint64_t)sub * dsub;
    uint32_t best = 0;
    float bd = INFINITY;
    for (uint32_t c = 0; c < ksub; ++c) {
        const float d = pq_dist(v, cent + (uint64_t)c * dsub, dsub);
        if (d < bd) {
            bd = d;
            best = c;
        }
    }
    if (assign) assign[r] = best;
    if (codes) codes[r * m + sub] = (uint8_t)best;
    if (any_nonzero && best != 0) *any_nonzero = 1u;  // `changed` (:279-281 against the zeroed assignments of :264)
}

// centroid update (:286-309): for (centroid c, dim d): sequential f32 sum over the rows assigned to c,
// in row order, divided by the count; untouched when no row is assigned.
__global__ void pq_update_kernel(const float* __restrict__ rows, uint64_t n, uint32_t dim, uint32_t sub, uint32_t dsub,
                                 const uint32_t* __restrict__ assign, float* __restrict__ cent, uint32_t ksub) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ksub * dsub) return;
    const uint32_t c = t / dsub, d = t % dsub;
    float s = 0.0f;
    uint32_t cnt = 0;
    const float* col = rows + (uint64_t)sub * dsub + d;
    for (uint64_t r = 0; r < n; ++r)
        if (assign[r] == c) {
            s = s + col[r * dim];
            ++cnt;
        }
    if (cnt) cent[(uint64_t)c * dsub + d] = s / (float)cnt;
}

__global__ void pq_init_kernel(const float* __restrict__ rows, uint64_t n, uint32_t dim, uint32_t sub, uint32_t dsub,
                               float* __restrict__ cent, uint32_t ksub) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ksub * dsub) return;
    const uint32_t c = t / dsub, d = t % dsub;
    cent[t] = rows[(uint64_t)(c % n) * dim + (uint64_t)sub * dsub + d];  // :255-258
}

__global__ void pq_decode_kernel(const uint8_t* __restrict__ codes, uint64_t n, uint32_t m, uint32_t dsub, uint32_t ksub,
                                 const float* __restrict__ cent /* [m][ksub][dsub] */, float* __restrict__ out,
                                 uint32_t* __restrict__ bad) {
    const uint64_t total = n * m * dsub;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ((uint64_t)m * dsub);
        const uint32_t rem = (uint32_t)(i % ((uint64_t)m * dsub)), sub = rem / dsub, d = rem % dsub;
        const uint32_t c = codes[r * m + sub];
        if (c >= ksub) {
            *bad = 1u;
            out[i] = 0.0f;
        } else {
            out[i] = cent[((uint64_t)sub * ksub + c) * dsub + d];
        }
    }
}

unsigned grid_for(uint64_t total) { return (unsigned)std::min<uint64_t>(4096, (total + 255) / 256); }

int need_gpu(int device_id) {
    if (cgv_device_count() == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    if (device_id < 0 || device_id >= cgv_device_count()) return fail(CGV_ERR_INVALID_ARG, "device_id out of range");
    return CGV_OK;
}

}  // namespace

struct cgv_sq {
    int device = 0;
    uint32_t dim = 0, nbits = 8;
    bool uniform = false, trained = false;
    std::vector<float> scales, biases;
    std::mutex mu;
};

struct cgv_pq {
    int device = 0;
    uint32_t dim = 0, m = 0, dsub = 0, nbits = 0, ksub = 0;
    bool trained = false;
    std::vector<float> cent;  // [m][ksub][dsub]
    std::mutex mu;
};

extern "C" {

int cgv_sq_create(uint32_t dim, uint32_t nbits, int uniform, int device_id, cgv_sq** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dim == 0 || nbits == 0 || nbits > 30) return fail(CGV_ERR_INVALID_ARG, "dim must be > 0 and nbits in 1..=30");
    int rc = need_gpu(device_id);
    if (rc) return rc;
    cgv_sq* q = new cgv_sq();
    q->device = device_id;
    q->dim = dim;
    q->nbits = nbits;
    q->uniform = uniform != 0;
    q->scales.assign(dim, 1.0f);  // :346-352
    q->biases.assign(dim, 0.0f);
    *out = q;
    return CGV_OK;
}

int cgv_sq_destroy(cgv_sq* q) {
    delete q;
    return CGV_OK;
}

uint32_t cgv_sq_bytes_per_value(const cgv_sq* q) { return !q ? 0u : (q->nbits == 8 ? 1u : (q->nbits == 16 ? 2u : 4u)); }

int cgv_sq_train_f32(cgv_sq* q, const float* rows_host, uint64_t n) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    if (n == 0) return fail(CGV_ERR_INVALID_ARG, "Cannot train on empty vector set");  // :356-360
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    QCHK(hipSetDevice(q->device));
    DevMem rows, mn, mx;
    QCHK(rows.alloc(n * q->dim * 4));
    QCHK(mn.alloc((size_t)q->dim * 4));
    QCHK(mx.alloc((size_t)q->dim * 4));
    QCHK(hipMemcpy(rows.p, rows_host, n * q->dim * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(col_minmax_kernel, dim3((q->dim + 63) / 64), dim3(64), 0, 0, rows.as<float>(), n, q->dim,
                       mn.as<float>(), mx.as<float>());
    QCHK(hipGetLastError());
    std::vector<float> lo(q->dim), hi(q->dim);
    QCHK(hipMemcpy(lo.data(), mn.p, (size_t)q->dim * 4, hipMemcpyDeviceToHost));
    QCHK(hipMemcpy(hi.data(), mx.p, (size_t)q->dim * 4, hipMemcpyDeviceToHost));
    q->scales.assign(q->dim, 1.0f);  // :363-364
    q->biases.assign(q->dim, 0.0f);
    const float levels = (float)(1 << q->nbits);
    if (q->uniform) {  // :366-385: one global range (min / max are order-free)
        float gmin = INFINITY, gmax = -INFINITY;
        for (uint32_t d = 0; d < q->dim; ++d) {
            if (lo[d] < gmin) gmin = lo[d];
            if (hi[d] > gmax) gmax = hi[d];
        }
        const float range = gmax - gmin;
        const float scale = levels / range;
        for (uint32_t d = 0; d < q->dim; ++d) {
            q->scales[d] = scale;
            q->biases[d] = gmin;
        }
    } else {  // :386-403
        for (uint32_t d = 0; d < q->dim; ++d) {
            const float range = hi[d] - lo[d];
            if (range > 0.0f) {
                q->scales[d] = levels / range;
                q->biases[d] = lo[d];
            }
        }
    }
    q->trained = true;
    return CGV_OK;
}

int cgv_sq_params(cgv_sq* q, float* scales_out, float* biases_out) {
    if (!q || !scales_out || !biases_out) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(q->mu);
    memcpy(scales_out, q->scales.data(), (size_t)q->dim * 4);
    memcpy(biases_out, q->biases.data(), (size_t)q->dim * 4);
    return CGV_OK;
}

static int sq_run(cgv_sq* q, const void* in_host, size_t in_bytes, void* out_host, size_t out_bytes, uint64_t n, bool enc) {
    QCHK(hipSetDevice(q->device));
    DevMem in, out, sc, bi;
    QCHK(in.alloc(in_bytes));
    QCHK(out.alloc(out_bytes));
    QCHK(sc.alloc((size_t)q->dim * 4));
    QCHK(bi.alloc((size_t)q->dim * 4));
    QCHK(hipMemcpy(in.p, in_host, in_bytes, hipMemcpyHostToDevice));
    QCHK(hipMemcpy(sc.p, q->scales.data(), (size_t)q->dim * 4, hipMemcpyHostToDevice));
    QCHK(hipMemcpy(bi.p, q->biases.data(), (size_t)q->dim * 4, hipMemcpyHostToDevice));
    if (enc)
        hipLaunchKernelGGL(sq_encode_kernel, dim3(grid_for(n * q->dim)), dim3(256), 0, 0, in.as<float>(), n, q->dim,
                           sc.as<float>(), bi.as<float>(), q->nbits, out.as<uint8_t>());
    else
        hipLaunchKernelGGL(sq_decode_kernel, dim3(grid_for(n * q->dim)), dim3(256), 0, 0, in.as<uint8_t>(), n, q->dim,
                           sc.as<float>(), bi.as<float>(), q->nbits, out.as<float>());
    QCHK(hipGetLastError());
    QCHK(hipMemcpy(out_host, out.p, out_bytes, hipMemcpyDeviceToHost));
    return CGV_OK;
}

int cgv_sq_encode_f32(cgv_sq* q, const float* rows_host, uint64_t n, uint8_t* out_host) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    if (!q->trained) return fail(CGV_ERR_INVALID_ARG, "Quantizer not trained");  // :411-413
    if (n == 0) return CGV_OK;
    if (!rows_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    const size_t bpv = q->nbits == 8 ? 1 : (q->nbits == 16 ? 2 : 4);
    return sq_run(q, rows_host, n * q->dim * 4, out_host, n * q->dim * bpv, n, true);
}

int cgv_sq_decode(cgv_sq* q, const uint8_t* codes_host, uint64_t n, float* out_host) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    if (!q->trained) return fail(CGV_ERR_INVALID_ARG, "Quantizer not trained");  // :439-441
    if (n == 0) return CGV_OK;
    if (!codes_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    const size_t bpv = q->nbits == 8 ? 1 : (q->nbits == 16 ? 2 : 4);
    return sq_run(q, codes_host, n * q->dim * bpv, out_host, n * q->dim * 4, n, false);
}

int cgv_pq_create(uint32_t dim, uint32_t m, uint32_t nbits, int device_id, cgv_pq** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dim == 0 || m == 0) return fail(CGV_ERR_INVALID_ARG, "dim and m must be > 0");
    if (dim % m != 0)  // :135-139
        return fail(CGV_ERR_INVALID_ARG, "Dimension must be divisible by number of subquantizers");
    if (nbits == 0 || nbits > 8) return fail(CGV_ERR_INVALID_ARG, "nbits must be in 1..=8 (codes are one byte, :214)");
    int rc = need_gpu(device_id);
    if (rc) return rc;
    cgv_pq* q = new cgv_pq();
    q->device = device_id;
    q->dim = dim;
    q->m = m;
    q->dsub = dim / m;
    q->nbits = nbits;
    q->ksub = 1u << nbits;
    q->cent.assign((size_t)m * q->ksub * q->dsub, 0.0f);  // :148
    *out = q;
    return CGV_OK;
}

int cgv_pq_destroy(cgv_pq* q) {
    delete q;
    return CGV_OK;
}

int cgv_pq_train_f32(cgv_pq* q, const float* rows_host, uint64_t n) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    if (n == 0) return fail(CGV_ERR_INVALID_ARG, "Cannot train on empty vector set");  // :156-160
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    QCHK(hipSetDevice(q->device));
    DevMem rows, cent, assign, flag;
    QCHK(rows.alloc(n * q->dim * 4));
    QCHK(cent.alloc(q->cent.size() * 4));
    QCHK(assign.alloc(n * 4));
    QCHK(flag.alloc(4));
    QCHK(hipMemcpy(rows.p, rows_host, n * q->dim * 4, hipMemcpyHostToDevice));
    const uint32_t kd = q->ksub * q->dsub;
    for (uint32_t sub = 0; sub < q->m; ++sub) {  // :168-181: sub-quantisers are independent
        float* c = cent.as<float>() + (size_t)sub * kd;
        hipLaunchKernelGGL(pq_init_kernel, dim3((kd + 255) / 256), dim3(256), 0, 0, rows.as<float>(), n, q->dim, sub, q->dsub,
                           c, q->ksub);
        for (int it = 0; it < 50; ++it) {  // :261
            QCHK(hipMemsetAsync(flag.p, 0, 4, 0));
            hipLaunchKernelGGL(pq_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, rows.as<float>(), n,
                               q->dim, sub, q->dsub, (const float*)c, q->ksub, assign.as<uint32_t>(), (uint8_t*)nullptr,
                               q->m, flag.as<uint32_t>());
            hipLaunchKernelGGL(pq_update_kernel, dim3((kd + 63) / 64), dim3(64), 0, 0, rows.as<float>(), n, q->dim, sub,
                               q->dsub, (const uint32_t*)assign.as<uint32_t>(), c, q->ksub);
            QCHK(hipGetLastError());
            uint32_t changed = 0;
            QCHK(hipMemcpy(&changed, flag.p, 4, hipMemcpyDeviceToHost));
            if (!changed) break;  // :311-313
        }
    }
    QCHK(hipMemcpy(q->cent.data(), cent.p, q->cent.size() * 4, hipMemcpyDeviceToHost));
    q->trained = true;
    return CGV_OK;
}

int cgv_pq_centroids(cgv_pq* q, float* out_host) {
    if (!q || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(q->mu);
    memcpy(out_host, q->cent.data(), q->cent.size() * 4);
    return CGV_OK;
}

int cgv_pq_encode_f32(cgv_pq* q, const float* rows_host, uint64_t n, uint8_t* codes_out) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    if (!q->trained) return fail(CGV_ERR_INVALID_ARG, "Quantizer not trained");  // :191-193
    if (n == 0) return CGV_OK;
    if (!rows_host || !codes_out) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    QCHK(hipSetDevice(q->device));
    DevMem rows, cent, codes;
    QCHK(rows.alloc(n * q->dim * 4));
    QCHK(cent.alloc(q->cent.size() * 4));
    QCHK(codes.alloc(n * q->m));
    QCHK(hipMemcpy(rows.p, rows_host, n * q->dim * 4, hipMemcpyHostToDevice));
    QCHK(hipMemcpy(cent.p, q->cent.data(), q->cent.size() * 4, hipMemcpyHostToDevice));
    const uint32_t kd = q->ksub * q->dsub;
    for (uint32_t sub = 0; sub < q->m; ++sub)
        hipLaunchKernelGGL(pq_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, rows.as<float>(), n, q->dim,
                           sub, q->dsub, (const float*)(cent.as<float>() + (size_t)sub * kd), q->ksub, (uint32_t*)nullptr,
                           codes.as<uint8_t>(), q->m, (uint32_t*)nullptr);
    QCHK(hipGetLastError());
    QCHK(hipMemcpy(codes_out, codes.p, n * q->m, hipMemcpyDeviceToHost));
    return CGV_OK;
}

int cgv_pq_decode(cgv_pq* q, const uint8_t* codes_host, uint64_t n, float* out_host) {
    if (!q) return fail(CGV_ERR_INVALID_ARG, "quantizer is NULL");
    std::lock_guard<std::mutex> lk(q->mu);
    if (!q->trained) return fail(CGV_ERR_INVALID_ARG, "Quantizer not trained");  // :222-224
    if (n == 0) return CGV_OK;
    if (!codes_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    QCHK(hipSetDevice(q->device));
    DevMem codes, cent, out, bad;
    QCHK(codes.alloc(n * q->m));
    QCHK(cent.alloc(q->cent.size() * 4));
    QCHK(out.alloc(n * q->dim * 4));
    QCHK(bad.alloc(4));
    QCHK(hipMemset(bad.p, 0, 4));
    QCHK(hipMemcpy(codes.p, codes_host, n * q->m, hipMemcpyHostToDevice));
    QCHK(hipMemcpy(cent.p, q->cent.data(), q->cent.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pq_decode_kernel, dim3(grid_for(n * q->dim)), dim3(256), 0, 0, codes.as<uint8_t>(), n, q->m, q->dsub,
                       q->ksub, cent.as<float>(), out.as<float>(), bad.as<uint32_t>());
    QCHK(hipGetLastError());
    uint32_t b = 0;
    QCHK(hipMemcpy(&b, bad.p, 4, hipMemcpyDeviceToHost));
    if (b) return fail(CGV_ERR_OUT_OF_RANGE, "Invalid centroid index");  // :236-238
    QCHK(hipMemcpy(out_host, out.p, n * q->dim * 4, hipMemcpyDeviceToHost));
    return CGV_OK;
}

}  // extern "C"
