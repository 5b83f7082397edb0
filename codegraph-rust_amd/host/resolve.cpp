// resolve.cpp — host-side mirror of the embedding phase of the reference's index-time symbol
// resolver, over the HIP kNN.
//
// Reference: crates/codegraph-mcp/src/indexer.rs
//   ai_semantic_match_sync, PHASE 2            :2790-2843  for one unresolved symbol: among the known
//       symbols that pass a cheap name filter (byte-length ratio >= 0.5 and character-trigram
//       Jaccard overlap >= 0.2 of the lower-cased names), the one with the highest
//       cosine_similarity_static to the unresolved symbol's embedding, if that exceeds 0.75;
//   cosine_similarity_static                   :2965-2979  sequential dot / (sqrt(na) * sqrt(nb))
//   char_trigrams / jaccard                    :2901-2932
// The reference walks a HashMap (iteration order unspecified) and keeps the first strictly greater
// similarity; this mirror defines ties as "lowest symbol index".
//
// Shape here: |unresolved| x |known| similarities are one batched search on the device
// (CGV_METRIC_COSINE_SEQ: MFMA coarse pass + exact sequential-cosine re-score, or the exact scan for
// an f32 index); the name filter — the reference's pre-filter, which also decides eligibility — is
// evaluated lazily on the host, only for the few best-scoring candidates of each target, walking
// down the exact ranking until an eligible symbol or the threshold is reached.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/cgvec.h"
#include "../../include/cgvec_store.h"

extern "C" int cgv_set_error_(int code, const char* msg);

namespace {

int fail(int code, const std::string& m) { return cgv_set_error_(code, m.c_str()); }

// str::to_lowercase restricted to ASCII (identifiers; non-ASCII letters are left as they are)
std::string lower_ascii(const char* s) {
    std::string r(s ? s : "");
    for (char& c : r)
        if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return r;
}

// byte offsets of the Unicode scalar values of a UTF-8 string (str::chars)
std::vector<size_t> char_starts(const std::string& s) {
    std::vector<size_t> st;
    for (size_t i = 0; i < s.size(); ++i)
        if (((unsigned char)s[i] & 0xC0) != 0x80) st.push_back(i);
    st.push_back(s.size());
    return st;
}

// indexer.rs:2901-2915: the set of 3-character windows; a string shorter than 3 characters is its
// own single element (nothing for the empty string)
std::unordered_set<std::string> char_trigrams(const std::string& s) {
    std::unordered_set<std::string> set;
    const std::vector<size_t> st = char_starts(s);
    const size_t nchars = st.size() - 1;
    if (nchars < 3) {
        if (!s.empty()) set.insert(s);
        return set;
    }
    for (size_t i = 0; i + 3 <= nchars; ++i) set.insert(s.substr(st[i], st[i + 3] - st[i]));
    return set;
}

// indexer.rs:2918-2932
float jaccard(const std::unordered_set<std::string>& a, const std::unordered_set<std::string>& b) {
    if (a.empty() || b.empty()) return 0.0f;
    size_t inter = 0;
    for (const std::string& t : a) inter += b.count(t);
    const float fi = (float)inter;
    const float uni = (float)(a.size() + b.size()) - fi;
    return uni == 0.0f ? 0.0f : fi / uni;
}

// the filter closure of indexer.rs:2804-2821
bool name_eligible(const std::string& target_lower, const std::unordered_set<std::string>& target_tri,
                   const std::string& name_lower) {
    const float a = (float)target_lower.size(), b = (float)name_lower.size();
    const float r1 = a / b, r2 = b / a;
    const float ratio = r1 < r2 ? r1 : r2;  // f32::min; NaN (0/0) compares false below
    if (!(ratio >= 0.5f)) return false;
    return jaccard(target_tri, char_trigrams(name_lower)) >= 0.2f;
}

}  // namespace

struct cgvs_resolver {
    cgv_index* index = nullptr;
    uint32_t dim = 0;
    std::vector<std::string> names_lower;
};

extern "C" {

int cgvs_resolver_create(uint32_t dim, int dtype, int device_id, cgvs_resolver** out) {
    if (!out) return fail(CGV_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    cgv_index* ix = nullptr;
    int rc = cgv_create(dim, CGV_METRIC_COSINE_SEQ, dtype, device_id, &ix);
    if (rc) return rc;
    cgvs_resolver* r = new cgvs_resolver();
    r->index = ix;
    r->dim = dim;
    *out = r;
    return CGV_OK;
}

int cgvs_resolver_destroy(cgvs_resolver* r) {
    if (!r) return CGV_OK;
    cgv_destroy(r->index);
    delete r;
    return CGV_OK;
}

int cgvs_resolver_add_symbols(cgvs_resolver* r, uint32_t n, const char* const* names, const float* embeddings) {
    if (!r) return fail(CGV_ERR_INVALID_ARG, "resolver is NULL");
    if (n == 0) return CGV_OK;
    if (!names || !embeddings) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    int rc = cgv_add_f32(r->index, embeddings, n);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; ++i) r->names_lower.push_back(lower_ascii(names[i]));
    return CGV_OK;
}

uint64_t cgvs_resolver_count(const cgvs_resolver* r) { return r ? r->names_lower.size() : 0; }

int cgvs_resolver_match(cgvs_resolver* r, uint32_t nq, const char* const* targets, const float* target_embeddings,
                        float threshold, int64_t* out_index, float* out_score) {
    if (!r) return fail(CGV_ERR_INVALID_ARG, "resolver is NULL");
    if (nq == 0) return CGV_OK;
    if (!targets || !target_embeddings || !out_index) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    const uint64_t n = r->names_lower.size();
    for (uint32_t q = 0; q < nq; ++q) {
        out_index[q] = -1;
        if (out_score) out_score[q] = 0.0f;
    }
    if (n == 0) return CGV_OK;
    const uint32_t K1 = (uint32_t)std::min<uint64_t>(32, n);
    std::vector<uint64_t> idx((size_t)nq * K1);
    std::vector<float> sc((size_t)nq * K1);
    int rc = cgv_search_f32(r->index, target_embeddings, nq, K1, idx.data(), sc.data());
    if (rc) return rc;
    std::vector<uint64_t> idx2;
    std::vector<float> sc2, all;
    for (uint32_t q = 0; q < nq; ++q) {
        const std::string tl = lower_ascii(targets[q]);
        const std::unordered_set<std::string> tri = char_trigrams(tl);
        // walk an exact ranking (score desc, index asc): first eligible symbol above the threshold wins
        auto walk = [&](const uint64_t* ii, const float* ss, uint32_t k, bool* exhausted) {
            *exhausted = false;
            for (uint32_t j = 0; j < k; ++j) {
                if (ii[j] == UINT64_MAX || !(ss[j] > threshold)) return;  // similarity > ai_threshold (:2828)
                if (name_eligible(tl, tri, r->names_lower[ii[j]])) {
                    out_index[q] = (int64_t)ii[j];
                    if (out_score) out_score[q] = ss[j];
                    return;
                }
            }
            *exhausted = (uint64_t)k < n;  // every listed symbol is above the threshold but none eligible
        };
        bool ex = false;
        walk(&idx[(size_t)q * K1], &sc[(size_t)q * K1], K1, &ex);
        if (!ex) continue;
        const uint32_t K2 = (uint32_t)std::min<uint64_t>(256, n);
        idx2.resize(K2);
        sc2.resize(K2);
        if ((rc = cgv_search_f32(r->index, target_embeddings + (size_t)q * r->dim, 1, K2, idx2.data(), sc2.data())))
            return rc;
        walk(idx2.data(), sc2.data(), K2, &ex);
        if (!ex) continue;
        // more than 256 ineligible symbols above the threshold: exact scores of every symbol
        all.resize(n);
        if ((rc = cgv_batch_similarity_f32(r->index, target_embeddings + (size_t)q * r->dim, CGV_OP_COSINE_SEQ, 0,
                                           all.data())))
            return rc;
        float best = threshold;
        for (uint64_t i = 0; i < n; ++i)
            if (all[i] > best && name_eligible(tl, tri, r->names_lower[i])) {  // strictly greater: lowest index on ties
                best = all[i];
                out_index[q] = (int64_t)i;
            }
        if (out_index[q] >= 0 && out_score) out_score[q] = best;
    }
    return CGV_OK;
}

// EmbeddingReRanker::rerank after the embeddings exist (crates/codegraph-vector/src/reranker.rs:113-157):
// cosine_similarity (:94-109, the sequential formula) of the query against every candidate, then a
// stable sort by score descending (ties keep the candidate order). Similarities on the device
// (f32 rows, exact scan); the <= ~100-element sort on the host.
int cgvs_rerank_embeddings(int device_id, const float* query, const float* candidates, uint32_t n, uint32_t dim,
                           uint32_t* out_order, float* out_score) {
    if (n == 0) return CGV_OK;
    if (!query || !candidates || !out_order || !out_score || dim == 0) return fail(CGV_ERR_INVALID_ARG, "NULL argument");
    cgv_index* ix = nullptr;
    int rc = cgv_create(dim, CGV_METRIC_COSINE_SEQ, CGV_DTYPE_F32, device_id, &ix);
    if (rc) return rc;
    std::vector<float> sc(n);
    rc = cgv_add_f32(ix, candidates, n);
    if (rc == CGV_OK) rc = cgv_batch_similarity_f32(ix, query, CGV_OP_COSINE_SEQ, 0, sc.data());
    cgv_destroy(ix);
    if (rc) return rc;
    std::vector<uint32_t> ord(n);
    for (uint32_t i = 0; i < n; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return sc[x] > sc[y]; });  // :148
    for (uint32_t i = 0; i < n; ++i) {
        out_order[i] = ord[i];
        out_score[i] = sc[ord[i]];
    }
    return CGV_OK;
}

float cgvs_trigram_jaccard(const char* a, const char* b) {
    return jaccard(char_trigrams(lower_ascii(a)), char_trigrams(lower_ascii(b)));
}

int cgvs_symbol_name_eligible(const char* target, const char* name) {
    const std::string tl = lower_ascii(target);
    return name_eligible(tl, char_trigrams(tl), lower_ascii(name)) ? 1 : 0;
}

}  // extern "C"
