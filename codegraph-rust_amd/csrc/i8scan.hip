// i8scan.hip — the reference's int8 "optimized" scan on the GPU (include/cgvec_i8.h).
//
// optimization.rs:63-150 scores one quantised query against N u8 rows with integer arithmetic;
// the scan is HBM-bound (N*D bytes read once, 8 B/row written back), so the kernel is a plain
// coalesced stream: 16 lanes per row, 16 bytes per lane per step, v_dot4_i32_i8 for both the dot
// product and |v|^2, a DPP-free shuffle reduction over the 16 lanes (integer: order-free, exact),
// and the reference's f32 score formula evaluated once per row with IEEE sqrt / divide.
// A second kernel keeps only the rows that can enter the reference's running buffer at all
// (score above the limit-th best of the first HEAD rows); the host replays the buffer policy
// (stable sorts, replace-the-minimum) over that short, index-ordered list.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/cgvec.h"
#include "../../include/cgvec_i8.h"

extern "C" int cgv_set_error_(int code, const char* msg);  // abi.hip (shared thread-local message)

namespace {

int fail(int code, const std::string& m) { return cgv_set_error_(code, m.c_str()); }

#define I8CHK(expr)                                                                             \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(_e == hipErrorOutOfMemory ? CGV_ERR_OOM : CGV_ERR_HIP,                  \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                     \
    } while (0)

// round(clamp(x,-1,1) * 127) -> clamp -> + 128   (optimization.rs:212-224, :270-271)
__host__ __device__ inline uint8_t quant_u8(float v) {
    float c = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);  // NaN stays NaN
    int q = (c != c) ? 0 : (int)roundf(c * 127.0f);       // Rust: NaN as i32 == 0
    q = q < -127 ? -127 : (q > 127 ? 127 : q);
    return (uint8_t)(q + 128);
}

__global__ void quantize_kernel(const float* __restrict__ in, uint64_t total, uint8_t* __restrict__ out) {
    uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < total) {
        const float4 v = *(const float4*)(in + i);
        uchar4 o;
        o.x = quant_u8(v.x);
        o.y = quant_u8(v.y);
        o.z = quant_u8(v.z);
        o.w = quant_u8(v.w);
        *(uchar4*)(out + i) = o;
    } else {
        for (; i < total; ++i) out[i] = quant_u8(in[i]);
    }
}

// quantize_unit_range_u4 (optimization.rs:338-343): round((clamp(x,-1,1)+1)/2 * 15) -> 0..15
__host__ __device__ inline uint8_t quant_u4(float v) {
    float c = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
    const float nrm = (c + 1.0f) / 2.0f;
    int q = (nrm != nrm) ? 0 : (int)roundf(nrm * 15.0f);  // Rust: NaN as i32 == 0
    q = q < 0 ? 0 : (q > 15 ? 15 : q);
    return (uint8_t)q;
}

// 4-bit arm of quantize_batch (optimization.rs:248-262): two values per byte, low nibble first,
// an odd last column pairs with 0
__global__ void quantize4_kernel(const float* __restrict__ in, uint64_t n, uint32_t dim, uint8_t* __restrict__ out) {
    const uint32_t half = (dim + 1) / 2;
    const uint64_t total = n * half;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / half;
        const uint32_t j = (uint32_t)(i % half) * 2;
        const uint8_t q0 = quant_u4(in[r * dim + j]);
        const uint8_t q1 = (j + 1 < dim) ? quant_u4(in[r * dim + j + 1]) : 0;
        out[i] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
    }
}

__device__ inline int sdot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// scores[row] = dot / (nq * sqrt(nv)), NaN when nv == 0 (row skipped by the reference, :131-133).
// q8: the quantised query as i8, padded with zeros to a multiple of 16 bytes.
__global__ __launch_bounds__(256) void i8_scores_kernel(const uint8_t* __restrict__ data, uint64_t n, uint32_t dim,
                                                       const int8_t* __restrict__ q8, float norm_query,
                                                       float* __restrict__ scores) {
    const int sub = threadIdx.x & 15;                                        // lane within the row group
    const uint64_t row = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);     // 16 rows per block
    int dot = 0, nv = 0;
    if (row < n) {
        const uint8_t* r = data + row * (uint64_t)dim;
        const uint32_t full = dim & ~15u;
        const bool aligned = ((uintptr_t)r & 15u) == 0;
        for (uint32_t j = sub * 16; j < full; j += 256) {
            uint4 v;
            if (aligned)
                v = *(const uint4*)(r + j);
            else
                memcpy(&v, r + j, 16);
            const uint4 q = *(const uint4*)(q8 + j);
            // u8 - 128 as i8  ==  u8 ^ 0x80
            const uint32_t a0 = v.x ^ 0x80808080u, a1 = v.y ^ 0x80808080u, a2 = v.z ^ 0x80808080u,
                           a3 = v.w ^ 0x80808080u;
            dot = sdot4(a0, q.x, dot);
            dot = sdot4(a1, q.y, dot);
            dot = sdot4(a2, q.z, dot);
            dot = sdot4(a3, q.w, dot);
            nv = sdot4(a0, a0, nv);
            nv = sdot4(a1, a1, nv);
            nv = sdot4(a2, a2, nv);
            nv = sdot4(a3, a3, nv);
        }
        if (sub == 0)
            for (uint32_t j = full; j < dim; ++j) {
                const int v = (int)r[j] - 128;
                dot += v * (int)q8[j];
                nv += v * v;
            }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        dot += __shfl_down(dot, off, 16);
        nv += __shfl_down(nv, off, 16);
    }
    if (row < n && sub == 0) scores[row] = (nv == 0) ? NAN : (float)dot / (norm_query * sqrtf((float)nv));
}

// Rows past `lo` whose score exceeds thr, appended in ANY order (the host sorts the short list by
// row index before replaying the buffer policy).
__global__ __launch_bounds__(256) void i8_filter_kernel(const float* __restrict__ scores, uint64_t lo, uint64_t n,
                                                       float thr, uint64_t* __restrict__ out_idx,
                                                       float* __restrict__ out_score, uint32_t cap,
                                                       uint32_t* __restrict__ out_cnt) {
    for (uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const float s = scores[i];
        if (s > thr) {  // false for NaN (skipped rows)
            const uint32_t p = atomicAdd(out_cnt, 1u);
            if (p < cap) {
                out_idx[p] = i;
                out_score[p] = s;
            }
        }
    }
}

}  // namespace

struct cgv_i8_index {
    int device = 0;
    uint32_t dim = 0;
    uint64_t n = 0, cap = 0;
    uint8_t* data = nullptr;  // [cap][dim] u8, row-major: the reference's optimized_data
    float* scores = nullptr;  // [cap]
    int8_t* q8 = nullptr;     // quantised query, dim rounded up to 16
    void* stage = nullptr;    // ingest staging
    size_t stage_bytes = 0;
    uint64_t* f_idx = nullptr;
    float* f_score = nullptr;
    uint32_t* f_cnt = nullptr;
    hipStream_t stream = nullptr;
    std::mutex mu;
};

namespace {

constexpr uint32_t FILTER_CAP = 1u << 20;
constexpr uint64_t HEAD = 8192;  // rows replayed unconditionally (their limit-th best is the filter threshold)

int grow(cgv_i8_index* h, uint64_t need) {
    if (need <= h->cap) return CGV_OK;
    uint64_t ncap = std::max<uint64_t>(need, h->cap + h->cap / 2);
    ncap = std::max<uint64_t>(ncap, 1024);
    uint8_t* d = nullptr;
    float* s = nullptr;
    I8CHK(hipMalloc((void**)&d, ncap * h->dim));
    I8CHK(hipMalloc((void**)&s, ncap * 4));
    if (h->n) I8CHK(hipMemcpyAsync(d, h->data, h->n * h->dim, hipMemcpyDeviceToDevice, h->stream));
    I8CHK(hipStreamSynchronize(h->stream));
    if (h->data) (void)hipFree(h->data);
    if (h->scores) (void)hipFree(h->scores);
    h->data = d;
    h->scores = s;
    h->cap = ncap;
    return CGV_OK;
}

int ensure_stage(cgv_i8_index* h, size_t bytes) {
    if (bytes <= h->stage_bytes) return CGV_OK;
    if (h->stage) (void)hipFree(h->stage);
    h->stage = nullptr;
    h->stage_bytes = 0;
    I8CHK(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return CGV_OK;
}

// Quantise the query like optimization.rs:86-96 and return |q| (:98-102), on the host: D values.
float prep_query(const float* query, uint32_t query_len, uint32_t dim, std::vector<int8_t>& q8) {
    q8.assign(((size_t)dim + 15) / 16 * 16, 0);
    for (uint32_t i = 0; i < dim && i < query_len; ++i) q8[i] = (int8_t)((int)quant_u8(query[i]) - 128);
    float nq = 0.0f;
    for (uint32_t i = 0; i < dim; ++i) nq += (float)q8[i] * (float)q8[i];  // sequential f32 sum, as the iterator
    return sqrtf(nq);
}

int launch_scores(cgv_i8_index* h, const float* query_host, uint32_t query_len, float* norm_out) {
    std::vector<int8_t> q8;
    const float nq = prep_query(query_host, query_len, h->dim, q8);
    *norm_out = nq;
    if (nq == 0.0f || h->n == 0) return CGV_OK;
    I8CHK(hipMemcpyAsync(h->q8, q8.data(), q8.size(), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(i8_scores_kernel, dim3((unsigned)((h->n + 15) / 16)), dim3(256), 0, h->stream, h->data, h->n,
                       h->dim, (const int8_t*)h->q8, nq, h->scores);
    I8CHK(hipGetLastError());
    return CGV_OK;
}

// The reference's running buffer (optimization.rs:117-149) over (idx, score) pairs in index order.
struct Replay {
    uint64_t limit;
    std::vector<std::pair<uint64_t, float>> best;
    static bool asc(const std::pair<uint64_t, float>& x, const std::pair<uint64_t, float>& y) {
        return x.second < y.second;
    }
    void push(uint64_t idx, float score) {
        if (best.size() < limit) {
            best.push_back({idx, score});
            if (best.size() == limit) std::stable_sort(best.begin(), best.end(), asc);
        } else if (score > best[0].second) {
            best[0] = {idx, score};
            std::stable_sort(best.begin(), best.end(), asc);
        }
    }
};

}  // namespace

extern "C" {

int cgv_i8_create(uint32_t dim, int device_id, cgv_i8_index** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (dim == 0 || dim > 8192) return fail(CGV_ERR_INVALID_ARG, "dim must be in 1..=8192");
    int ndev = cgv_device_count();
    if (ndev == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(CGV_ERR_INVALID_ARG, "device_id out of range");
    I8CHK(hipSetDevice(device_id));
    cgv_i8_index* h = new cgv_i8_index();
    h->device = device_id;
    h->dim = dim;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->q8, ((size_t)dim + 15) / 16 * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&h->f_idx, (size_t)FILTER_CAP * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&h->f_score, (size_t)FILTER_CAP * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->f_cnt, 4);
    if (e != hipSuccess) {
        std::string m = hipGetErrorString(e);
        cgv_i8_destroy(h);
        return fail(CGV_ERR_HIP, "cgv_i8_create: " + m);
    }
    *out = h;
    return CGV_OK;
}

int cgv_i8_destroy(cgv_i8_index* h) {
    if (!h) return CGV_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* bufs[] = {h->data, h->scores, h->q8, h->stage, h->f_idx, h->f_score, h->f_cnt};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return CGV_OK;
}

int cgv_i8_add_u8(cgv_i8_index* h, const uint8_t* data_host, uint64_t n) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n == 0) return CGV_OK;
    if (!data_host) return fail(CGV_ERR_INVALID_ARG, "data is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    I8CHK(hipSetDevice(h->device));
    int rc = grow(h, h->n + n);
    if (rc) return rc;
    I8CHK(hipMemcpyAsync(h->data + h->n * h->dim, data_host, n * h->dim, hipMemcpyHostToDevice, h->stream));
    I8CHK(hipStreamSynchronize(h->stream));
    h->n += n;
    return CGV_OK;
}

int cgv_i8_add_f32(cgv_i8_index* h, const float* rows_host, uint64_t n) {
    if (!h) return fail(CGV_ERR_INVALID_ARG, "handle is NULL");
    if (n == 0) return CGV_OK;
    if (!rows_host) return fail(CGV_ERR_INVALID_ARG, "rows is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    I8CHK(hipSetDevice(h->device));
    int rc = grow(h, h->n + n);
    if (rc) return rc;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / ((uint64_t)h->dim * 4));
    for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
        const uint64_t c = std::min<uint64_t>(chunk, n - r0);
        const uint64_t total = c * h->dim;
        if ((rc = ensure_stage(h, total * 4))) return rc;
        I8CHK(hipMemcpyAsync(h->stage, rows_host + r0 * h->dim, total * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((total / 4 + 255) / 256 + 1)), dim3(256), 0, h->stream,
                           (const float*)h->stage, total, h->data + (h->n + r0) * h->dim);
        I8CHK(hipGetLastError());
        I8CHK(hipStreamSynchronize(h->stream));
    }
    h->n += n;
    return CGV_OK;
}

uint64_t cgv_i8_count(const cgv_i8_index* h) { return h ? h->n : 0; }

int cgv_i8_get_row_u8(cgv_i8_index* h, uint64_t id, uint8_t* out_host) {
    if (!h || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (id >= h->n) return fail(CGV_ERR_OUT_OF_RANGE, "row id out of range");
    I8CHK(hipSetDevice(h->device));
    I8CHK(hipMemcpyAsync(out_host, h->data + id * h->dim, h->dim, hipMemcpyDeviceToHost, h->stream));
    I8CHK(hipStreamSynchronize(h->stream));
    return CGV_OK;
}

int cgv_i8_scores_f32(cgv_i8_index* h, const float* query_host, uint32_t query_len, float* out_host) {
    if (!h || !query_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->n == 0) return CGV_OK;
    I8CHK(hipSetDevice(h->device));
    float nq = 0.0f;
    int rc = launch_scores(h, query_host, query_len, &nq);
    if (rc) return rc;
    if (nq == 0.0f) {
        for (uint64_t i = 0; i < h->n; ++i) out_host[i] = NAN;
        return CGV_OK;
    }
    I8CHK(hipMemcpyAsync(out_host, h->scores, h->n * 4, hipMemcpyDeviceToHost, h->stream));
    I8CHK(hipStreamSynchronize(h->stream));
    return CGV_OK;
}

int cgv_i8_search_optimized(cgv_i8_index* h, const float* query_host, uint32_t query_len, uint64_t limit_in,
                            uint64_t* out_idx_host, uint64_t* out_n) {
    if (!h || !query_host || !out_idx_host || !out_n) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    *out_n = 0;
    const uint64_t limit = std::max<uint64_t>(limit_in, 1);  // optimization.rs:64
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->n == 0) return CGV_OK;  // :71-73
    I8CHK(hipSetDevice(h->device));
    float nq = 0.0f;
    int rc = launch_scores(h, query_host, query_len, &nq);
    if (rc) return rc;
    if (nq == 0.0f) return CGV_OK;  // :113-115

    Replay rp;
    rp.limit = limit;
    rp.best.reserve((size_t)std::min<uint64_t>(limit, h->n));
    const uint64_t head = std::min<uint64_t>(h->n, std::max<uint64_t>(HEAD, 2 * limit));
    std::vector<float> hs(head);
    I8CHK(hipMemcpyAsync(hs.data(), h->scores, head * 4, hipMemcpyDeviceToHost, h->stream));
    I8CHK(hipStreamSynchronize(h->stream));
    for (uint64_t i = 0; i < head; ++i)
        if (hs[i] == hs[i]) rp.push(i, hs[i]);
    if (head < h->n) {
        // Past the head a row can enter the buffer only if its score exceeds the buffer's current
        // minimum, which never decreases: filter on the device with the minimum reached so far
        // (-inf while the buffer is not full), then replay the survivors in index order.
        const float thr = rp.best.size() == limit ? rp.best[0].second : -INFINITY;
        I8CHK(hipMemsetAsync(h->f_cnt, 0, 4, h->stream));
        const unsigned fb = (unsigned)std::min<uint64_t>(2048, (h->n - head + 255) / 256);
        hipLaunchKernelGGL(i8_filter_kernel, dim3(fb), dim3(256), 0, h->stream, (const float*)h->scores, head, h->n,
                           thr, h->f_idx, h->f_score, FILTER_CAP, h->f_cnt);
        I8CHK(hipGetLastError());
        uint32_t cnt = 0;
        I8CHK(hipMemcpyAsync(&cnt, h->f_cnt, 4, hipMemcpyDeviceToHost, h->stream));
        I8CHK(hipStreamSynchronize(h->stream));
        if (cnt <= FILTER_CAP) {
            std::vector<uint64_t> fi(cnt);
            std::vector<float> fs(cnt);
            if (cnt) {
                I8CHK(hipMemcpyAsync(fi.data(), h->f_idx, (size_t)cnt * 8, hipMemcpyDeviceToHost, h->stream));
                I8CHK(hipMemcpyAsync(fs.data(), h->f_score, (size_t)cnt * 4, hipMemcpyDeviceToHost, h->stream));
                I8CHK(hipStreamSynchronize(h->stream));
            }
            std::vector<uint32_t> ord(cnt);
            for (uint32_t j = 0; j < cnt; ++j) ord[j] = j;
            std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return fi[x] < fi[y]; });
            for (uint32_t j = 0; j < cnt; ++j) rp.push(fi[ord[j]], fs[ord[j]]);
        } else {  // pathological (more than 2^20 survivors): replay every score
            std::vector<float> all(h->n - head);
            I8CHK(hipMemcpyAsync(all.data(), h->scores + head, (h->n - head) * 4, hipMemcpyDeviceToHost, h->stream));
            I8CHK(hipStreamSynchronize(h->stream));
            for (uint64_t i = head; i < h->n; ++i)
                if (all[i - head] == all[i - head]) rp.push(i, all[i - head]);
        }
    }
    std::stable_sort(rp.best.begin(), rp.best.end(),
                     [](const std::pair<uint64_t, float>& x, const std::pair<uint64_t, float>& y) {
                         return x.second > y.second;
                     });  // :148
    for (size_t j = 0; j < rp.best.size(); ++j) out_idx_host[j] = rp.best[j].first;
    *out_n = rp.best.size();
    return CGV_OK;
}

int cgv_quantize_u8_f32(int device_id, const float* rows_host, uint64_t n, uint32_t dim, uint8_t* out_host) {
    if (n == 0 || dim == 0) return CGV_OK;
    if (!rows_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    if (cgv_device_count() == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    I8CHK(hipSetDevice(device_id));
    const uint64_t total = n * dim;
    float* din = nullptr;
    uint8_t* dout = nullptr;
    I8CHK(hipMalloc((void**)&din, total * 4));
    hipError_t e = hipMalloc((void**)&dout, total);
    if (e == hipSuccess) e = hipMemcpy(din, rows_host, total * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)((total / 4 + 255) / 256 + 1)), dim3(256), 0, 0,
                           (const float*)din, total, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_host, dout, total, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(CGV_ERR_HIP, std::string("cgv_quantize_u8_f32: ") + hipGetErrorString(e));
    return CGV_OK;
}

int cgv_quantize_u4_f32(int device_id, const float* rows_host, uint64_t n, uint32_t dim, uint8_t* out_host) {
    if (n == 0 || dim == 0) return CGV_OK;
    if (!rows_host || !out_host) return fail(CGV_ERR_INVALID_ARG, "NULL buffer");
    if (cgv_device_count() == 0) return fail(CGV_ERR_HIP, "no HIP device visible: libcgvec_hip has no CPU fallback");
    I8CHK(hipSetDevice(device_id));
    const uint64_t total = n * dim, obytes = n * ((dim + 1) / 2);
    float* din = nullptr;
    uint8_t* dout = nullptr;
    I8CHK(hipMalloc((void**)&din, total * 4));
    hipError_t e = hipMalloc((void**)&dout, obytes);
    if (e == hipSuccess) e = hipMemcpy(din, rows_host, total * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(quantize4_kernel, dim3((unsigned)std::min<uint64_t>(4096, (obytes + 255) / 256)), dim3(256), 0, 0,
                           (const float*)din, n, dim, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_host, dout, obytes, hipMemcpyDeviceToHost);
    (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(CGV_ERR_HIP, std::string("cgv_quantize_u4_f32: ") + hipGetErrorString(e));
    return CGV_OK;
}

}  // extern "C"
