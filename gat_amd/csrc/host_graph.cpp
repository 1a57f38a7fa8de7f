// Host-side integer work of libgatmi: COO -> CSR -> CSC, work-item lists, row partitions and
// halo plans.  Everything here is deterministic and bit-exact by construction (stable counting
// sorts, no floating point).  It picks up where the reference's preprocess_adj_bias
// (utils/process.py:179-188) stops: that function emits the (indices, values, shape) triple of
// A + I; these routines turn the triple into the structures the HIP kernels walk.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <utility>
#include <vector>

#include "common.h"

namespace gatmi {

static thread_local char g_err[512] = "";

char *error_buffer() { return g_err; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace gatmi

using gatmi::fail;

extern "C" int gatmi_version(void) { return 100; }

extern "C" const char *gatmi_last_error(void) { return gatmi::error_buffer(); }

extern "C" int gatmi_csr_from_coo(int64_t n, int64_t nnz, const int32_t *dst, const int32_t *src,
                                  int add_self_loops, int dedup, int64_t cap,
                                  int64_t *rowptr_out, int32_t *col_out, int64_t *perm_out,
                                  int64_t *nnz_out) {
    GATMI_REQUIRE(n >= 0 && nnz >= 0, GATMI_EINVAL, "csr_from_coo: negative size (n=%lld nnz=%lld)", (long long)n, (long long)nnz);
    GATMI_REQUIRE(rowptr_out && nnz_out && (col_out || cap == 0), GATMI_EINVAL, "csr_from_coo: null output");
    GATMI_REQUIRE(nnz == 0 || (dst && src), GATMI_EINVAL, "csr_from_coo: null input");
    GATMI_REQUIRE(n < INT32_MAX, GATMI_ERANGE, "csr_from_coo: n=%lld does not fit int32 ids", (long long)n);
    GATMI_REQUIRE(cap >= nnz + (add_self_loops ? n : 0), GATMI_EINVAL,
                  "csr_from_coo: cap=%lld < nnz + self loops = %lld", (long long)cap, (long long)(nnz + (add_self_loops ? n : 0)));
    for (int64_t k = 0; k < nnz; ++k) {
        if (dst[k] < 0 || dst[k] >= n || src[k] < 0 || src[k] >= n)
            return fail(GATMI_ERANGE, "csr_from_coo: entry %lld = (%d,%d) outside [0,%lld)", (long long)k, dst[k], src[k], (long long)n);
    }
    // stable counting sort by destination
    std::vector<int64_t> start(n + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) start[dst[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) start[i + 1] += start[i];
    std::vector<std::pair<int32_t, int64_t>> ent(nnz);  // (src, input position), grouped by dst
    {
        std::vector<int64_t> fill(start.begin(), start.end() - 1);
        for (int64_t k = 0; k < nnz; ++k) ent[fill[dst[k]]++] = {src[k], k};
    }
    int64_t w = 0;
    rowptr_out[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        auto b = ent.begin() + start[i], e = ent.begin() + start[i + 1];
        std::sort(b, e);  // by source id, ties by input position
        bool self_seen = false;
        bool self_pending = add_self_loops != 0;
        int32_t last = -1;
        for (auto it = b; it != e; ++it) {
            if (self_pending && !self_seen && it->first > (int32_t)i) {
                col_out[w] = (int32_t)i;
                if (perm_out) perm_out[w] = -1;
                ++w;
                self_seen = true;
            }
            if (dedup && it != b && it->first == last) continue;
            last = it->first;
            if (it->first == (int32_t)i) self_seen = true;
            col_out[w] = it->first;
            if (perm_out) perm_out[w] = it->second;
            ++w;
        }
        if (self_pending && !self_seen) {
            col_out[w] = (int32_t)i;
            if (perm_out) perm_out[w] = -1;
            ++w;
        }
        rowptr_out[i + 1] = w;
    }
    *nnz_out = w;
    return GATMI_OK;
}

extern "C" int gatmi_csc_from_csr(int64_t n_rows, int64_t n_cols, const int64_t *rowptr,
                                  const int32_t *col, int64_t *colptr_out, int32_t *rowidx_out,
                                  int32_t *perm_out) {
    GATMI_REQUIRE(n_rows >= 0 && n_cols >= 0 && rowptr && colptr_out, GATMI_EINVAL, "csc_from_csr: bad arguments");
    const int64_t nnz = rowptr[n_rows];
    GATMI_REQUIRE(nnz == 0 || (col && rowidx_out), GATMI_EINVAL, "csc_from_csr: null array");
    GATMI_REQUIRE(nnz < INT32_MAX && n_rows < INT32_MAX, GATMI_ERANGE, "csc_from_csr: %lld entries do not fit int32 offsets", (long long)nnz);
    std::fill(colptr_out, colptr_out + n_cols + 1, (int64_t)0);
    for (int64_t k = 0; k < nnz; ++k) {
        if (col[k] < 0 || col[k] >= n_cols) return fail(GATMI_ERANGE, "csc_from_csr: col[%lld]=%d outside [0,%lld)", (long long)k, col[k], (long long)n_cols);
        colptr_out[col[k] + 1]++;
    }
    for (int64_t j = 0; j < n_cols; ++j) colptr_out[j + 1] += colptr_out[j];
    std::vector<int64_t> fill(colptr_out, colptr_out + n_cols);
    for (int64_t i = 0; i < n_rows; ++i) {          // ascending i => each column sorted by i
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            int64_t p = fill[col[k]]++;
            rowidx_out[p] = (int32_t)i;
            if (perm_out) perm_out[p] = (int32_t)k;
        }
    }
    return GATMI_OK;
}

static inline int64_t pieces_of(int64_t len, int64_t chunk) { return len <= chunk ? 1 : (len + chunk - 1) / chunk; }

extern "C" int gatmi_work_items_count(int64_t n, const int64_t *ptr, int32_t chunk,
                                      int64_t *n_items_out, int64_t *n_split_out, int64_t *n_slots_out) {
    GATMI_REQUIRE(n >= 0 && ptr && chunk > 0 && n_items_out && n_split_out && n_slots_out, GATMI_EINVAL, "work_items_count: bad arguments");
    GATMI_REQUIRE(ptr[n] < INT32_MAX && n < INT32_MAX, GATMI_ERANGE, "work_items: %lld entries do not fit int32 offsets", (long long)ptr[n]);
    int64_t items = 0, split = 0, slots = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = ptr[i + 1] - ptr[i];
        if (len < 0) return fail(GATMI_EINVAL, "work_items: ptr not monotone at row %lld", (long long)i);
        int64_t p = pieces_of(len, chunk);
        items += p;
        if (p > 1) { split++; slots += p; }
    }
    *n_items_out = items;
    *n_split_out = split;
    *n_slots_out = slots;
    return GATMI_OK;
}

extern "C" int gatmi_work_items(int64_t n, const int64_t *ptr, int32_t chunk,
                                int32_t *items_out, int32_t *split_out) {
    int64_t n_items, n_split, n_slots;
    int rc = gatmi_work_items_count(n, ptr, chunk, &n_items, &n_split, &n_slots);
    if (rc) return rc;
    GATMI_REQUIRE(items_out || n_items == 0, GATMI_EINVAL, "work_items: null items_out");
    GATMI_REQUIRE(split_out || n_split == 0, GATMI_EINVAL, "work_items: null split_out");
    // counting sort of the pieces by length, longest first, stable by (row, piece)
    std::vector<int64_t> bucket(chunk + 2, 0);
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = ptr[i + 1] - ptr[i];
        if (len <= chunk) bucket[len]++;
        else {
            int64_t p = pieces_of(len, chunk);
            bucket[chunk] += p - 1;
            bucket[len - (p - 1) * chunk]++;
        }
    }
    std::vector<int64_t> pos(chunk + 2, 0);
    {
        int64_t acc = 0;
        for (int64_t l = chunk; l >= 0; --l) { pos[l] = acc; acc += bucket[l]; }
    }
    int64_t slot = 0, s = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = ptr[i + 1] - ptr[i];
        int64_t p = pieces_of(len, chunk);
        if (p == 1) {
            int32_t *it = items_out + 4 * pos[len]++;
            it[0] = (int32_t)i; it[1] = (int32_t)ptr[i]; it[2] = (int32_t)len; it[3] = -1;
        } else {
            split_out[4 * s + 0] = (int32_t)i;
            split_out[4 * s + 1] = (int32_t)slot;
            split_out[4 * s + 2] = (int32_t)p;
            split_out[4 * s + 3] = 0;
            ++s;
            for (int64_t c = 0; c < p; ++c) {
                int64_t b = ptr[i] + c * chunk;
                int64_t l = std::min<int64_t>(chunk, ptr[i + 1] - b);
                int32_t *it = items_out + 4 * pos[l]++;
                it[0] = (int32_t)i; it[1] = (int32_t)b; it[2] = (int32_t)l; it[3] = (int32_t)(slot + c);
            }
            slot += p;
        }
    }
    return GATMI_OK;
}

// ---- two-phase work items (row-partitioned shards) ---------------------------------------------
namespace {
struct PhasedCount { int64_t a = 0, b = 0, split = 0, slots = 0; };

inline int64_t pieces_or_none(int64_t len, int64_t chunk) { return len <= 0 ? 0 : pieces_of(len, chunk); }

int phased_count(int64_t n, const int64_t *ptr, const int64_t *cut, int32_t chunk, PhasedCount *c) {
    GATMI_REQUIRE(n >= 0 && ptr && cut && chunk > 0, GATMI_EINVAL, "work_items_phased: bad arguments");
    GATMI_REQUIRE(ptr[n] < INT32_MAX && n < INT32_MAX, GATMI_ERANGE, "work_items_phased: %lld entries do not fit int32 offsets", (long long)ptr[n]);
    for (int64_t i = 0; i < n; ++i) {
        if (cut[i] < ptr[i] || cut[i] > ptr[i + 1])
            return fail(GATMI_EINVAL, "work_items_phased: cut[%lld]=%lld outside its row [%lld,%lld]", (long long)i, (long long)cut[i], (long long)ptr[i], (long long)ptr[i + 1]);
        int64_t pa = pieces_or_none(cut[i] - ptr[i], chunk), pb = pieces_or_none(ptr[i + 1] - cut[i], chunk);
        if (pa + pb == 0) pa = 1;                       // an empty row still writes its output
        c->a += pa;
        c->b += pb;
        if (pa == 1 && pb == 1) c->slots += 1;          // parked by phase A, resumed by phase B
        else if (pa + pb > 1) { c->split++; c->slots += pa + pb; }
    }
    return GATMI_OK;
}

// positions of the pieces of one phase in its list: longest first, stable by (row, piece)
struct LengthOrder {
    std::vector<int64_t> pos;
    explicit LengthOrder(int32_t chunk) : pos(chunk + 2, 0) {}
    void count(int64_t len, int64_t chunk) {
        if (len <= chunk) pos[len]++;
        else {
            int64_t p = pieces_of(len, chunk);
            pos[chunk] += p - 1;
            pos[len - (p - 1) * chunk]++;
        }
    }
    void finish(int64_t chunk) {
        int64_t acc = 0;
        for (int64_t l = chunk; l >= 0; --l) { int64_t c = pos[l]; pos[l] = acc; acc += c; }
    }
};
}  // namespace

extern "C" int gatmi_work_items_phased_count(int64_t n, const int64_t *ptr, const int64_t *cut, int32_t chunk,
                                             int64_t *n_a_out, int64_t *n_b_out, int64_t *n_split_out, int64_t *n_slots_out) {
    GATMI_REQUIRE(n_a_out && n_b_out && n_split_out && n_slots_out, GATMI_EINVAL, "work_items_phased_count: null output");
    PhasedCount c;
    int rc = phased_count(n, ptr, cut, chunk, &c);
    if (rc) return rc;
    *n_a_out = c.a; *n_b_out = c.b; *n_split_out = c.split; *n_slots_out = c.slots;
    return GATMI_OK;
}

extern "C" int gatmi_work_items_phased(int64_t n, const int64_t *ptr, const int64_t *cut, int32_t chunk,
                                       int32_t *items_a_out, int32_t *items_b_out, int32_t *split_out) {
    PhasedCount c;
    int rc = phased_count(n, ptr, cut, chunk, &c);
    if (rc) return rc;
    GATMI_REQUIRE((items_a_out || c.a == 0) && (items_b_out || c.b == 0) && (split_out || c.split == 0), GATMI_EINVAL, "work_items_phased: null output");
    LengthOrder oa(chunk), ob(chunk);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t la = cut[i] - ptr[i], lb = ptr[i + 1] - cut[i];
        if (la > 0 || lb == 0) oa.count(la, chunk);
        if (lb > 0) ob.count(lb, chunk);
    }
    oa.finish(chunk);
    ob.finish(chunk);
    int64_t slot = 0, s = 0;
    auto emit = [&](int32_t *out, LengthOrder &o, int64_t row, int64_t first, int64_t len, int64_t w) {
        int32_t *it = out + 4 * o.pos[len]++;
        it[0] = (int32_t)row; it[1] = (int32_t)first; it[2] = (int32_t)len; it[3] = (int32_t)w;
    };
    for (int64_t i = 0; i < n; ++i) {
        const int64_t la = cut[i] - ptr[i], lb = ptr[i + 1] - cut[i];
        int64_t pa = pieces_or_none(la, chunk), pb = pieces_or_none(lb, chunk);
        if (pa + pb == 0) pa = 1;
        if (pa + pb == 1) {
            if (pa) emit(items_a_out, oa, i, ptr[i], la, -1);
            else emit(items_b_out, ob, i, cut[i], lb, -1);
        } else if (pa == 1 && pb == 1) {
            emit(items_a_out, oa, i, ptr[i], la, slot);
            emit(items_b_out, ob, i, cut[i], lb, -(slot + 2));
            slot += 1;
        } else {
            split_out[4 * s + 0] = (int32_t)i;
            split_out[4 * s + 1] = (int32_t)slot;
            split_out[4 * s + 2] = (int32_t)(pa + pb);
            split_out[4 * s + 3] = 0;
            ++s;
            for (int64_t k = 0; k < pa; ++k) {
                const int64_t b = ptr[i] + k * chunk;
                emit(items_a_out, oa, i, b, std::min<int64_t>(chunk, cut[i] - b), slot + k);
            }
            for (int64_t k = 0; k < pb; ++k) {
                const int64_t b = cut[i] + k * chunk;
                emit(items_b_out, ob, i, b, std::min<int64_t>(chunk, ptr[i + 1] - b), slot + pa + k);
            }
            slot += pa + pb;
        }
    }
    return GATMI_OK;
}

extern "C" int gatmi_partition_rows(int64_t n, int32_t nparts, const int64_t *rowptr, int32_t balance,
                                    int64_t *part_out) {
    GATMI_REQUIRE(n >= 0 && nparts > 0 && part_out, GATMI_EINVAL, "partition_rows: bad arguments");
    GATMI_REQUIRE(balance == 0 || rowptr, GATMI_EINVAL, "partition_rows: rowptr needed for edge balance");
    part_out[0] = 0;
    if (balance == 0) {
        for (int32_t p = 1; p <= nparts; ++p) part_out[p] = (n * (int64_t)p) / nparts;
    } else {
        const int64_t nnz = rowptr[n];
        int64_t i = 0;
        for (int32_t p = 1; p < nparts; ++p) {
            // first row whose prefix reaches p/nparts of the entries (integer arithmetic only)
            const __int128 target = (__int128)nnz * p;
            while (i < n && (__int128)rowptr[i] * nparts < target) ++i;
            part_out[p] = i;
        }
        part_out[nparts] = n;
    }
    return GATMI_OK;
}

extern "C" int gatmi_halo_plan(int32_t nparts, int32_t rank, const int64_t *part, int64_t nnz,
                               const int64_t *col_global, int32_t *col_local_out,
                               int64_t *halo_ids_out, int64_t *halo_counts_out, int64_t *n_halo_out) {
    GATMI_REQUIRE(nparts > 0 && rank >= 0 && rank < nparts && part && n_halo_out && halo_counts_out, GATMI_EINVAL, "halo_plan: bad arguments");
    GATMI_REQUIRE(nnz == 0 || (col_global && col_local_out && halo_ids_out), GATMI_EINVAL, "halo_plan: null array");
    const int64_t lo = part[rank], hi = part[rank + 1], n_total = part[nparts];
    std::vector<int64_t> remote;
    remote.reserve(nnz / 2 + 1);
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t j = col_global[k];
        if (j < 0 || j >= n_total) return fail(GATMI_ERANGE, "halo_plan: col[%lld]=%lld outside [0,%lld)", (long long)k, (long long)j, (long long)n_total);
        if (j < lo || j >= hi) remote.push_back(j);
    }
    std::sort(remote.begin(), remote.end());
    remote.erase(std::unique(remote.begin(), remote.end()), remote.end());
    const int64_t n_halo = (int64_t)remote.size();
    GATMI_REQUIRE((hi - lo) + n_halo < INT32_MAX, GATMI_ERANGE, "halo_plan: %lld local+halo rows do not fit int32", (long long)((hi - lo) + n_halo));
    std::copy(remote.begin(), remote.end(), halo_ids_out);
    for (int32_t p = 0; p < nparts; ++p) {
        auto b = std::lower_bound(remote.begin(), remote.end(), part[p]);
        auto e = std::lower_bound(remote.begin(), remote.end(), part[p + 1]);
        halo_counts_out[p] = (p == rank) ? 0 : (int64_t)(e - b);
    }
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t j = col_global[k];
        if (j >= lo && j < hi) col_local_out[k] = (int32_t)(j - lo);
        else col_local_out[k] = (int32_t)((hi - lo) + (std::lower_bound(remote.begin(), remote.end(), j) - remote.begin()));
    }
    *n_halo_out = n_halo;
    return GATMI_OK;
}
