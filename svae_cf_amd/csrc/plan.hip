// Host-side batch-plan builder (no device code): what svae_cf_amd/batch.py:build_plan computes with numpy, as one
// C call that holds no interpreter lock -- the prefetcher's worker threads really run side by side, and a 256-user
// ML-20M-shaped plan costs a fraction of a millisecond instead of 2-3.
//
// Replaces, per batch of B users, the per-user packing of the reference's DataReader.iter (nb:485-526: x [1,T], the
// dense one-hot next-k targets y_s [1,T,I] built by T scatter_ calls, nb:509-513) -- see batch.py for the layout
// (time-major rows, users sorted by length, int32 target table, per-row loss weights, rows grouped by input item
// for the embedding gradient, target occurrences grouped by item for dL/dW3).  The numpy build_plan stays the
// specification: tests/test_host_logic.py checks the two byte for byte.
#include "common.h"
#include "../../include/svae_hip.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr int EMBED_LONG_ROWS = 16;   // = svae_embed_long_rows()
constexpr int OCC_LONG = 32;          // = svae_proj_occ_long()

struct Plan {
    svae_plan_info info;
    std::vector<int32_t> x, tgt, row_off, nact, last_row, perm, seg_off, seg_item, seen, seen_off, occ_off, occ_row,
        occ_items;
    std::vector<float> w_nll, w_kl;
    std::vector<int64_t> row_key, order, lengths;
};

// stable LSD radix sort of (key, payload) pairs, 11-bit digits, as many passes as max_key needs
void radix_sort_pairs(std::vector<uint32_t>& key, std::vector<uint32_t>& val, uint32_t max_key) {
    const size_t n = key.size();
    if (n < 2) return;
    std::vector<uint32_t> k2(n), v2(n);
    uint32_t cnt[2048];
    for (int shift = 0; shift < 32 && (max_key >> shift) != 0; shift += 11) {
        std::memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; ++i) ++cnt[(key[i] >> shift) & 2047u];
        uint32_t run = 0;
        for (int d = 0; d < 2048; ++d) {
            const uint32_t c = cnt[d];
            cnt[d] = run;
            run += c;
        }
        for (size_t i = 0; i < n; ++i) {
            const uint32_t p = cnt[(key[i] >> shift) & 2047u]++;
            k2[p] = key[i];
            v2[p] = val[i];
        }
        key.swap(k2);
        val.swap(v2);
    }
}

// indices 0..n-1 by count descending, ties by index ascending (counting sort; max_cnt = the largest count)
void order_by_count_desc(const int64_t* cnt, int64_t n, int64_t max_cnt, std::vector<int32_t>& out) {
    std::vector<int64_t> start((size_t)max_cnt + 2, 0);
    for (int64_t i = 0; i < n; ++i) ++start[(size_t)(max_cnt - cnt[i]) + 1];
    for (int64_t c = 0; c <= max_cnt; ++c) start[(size_t)c + 1] += start[(size_t)c];
    out.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) out[(size_t)start[(size_t)(max_cnt - cnt[i])]++] = (int32_t)i;
}

// batch.py:choose_time_split (same constants, same order of the floating-point operations)
void choose_time_split(const int32_t* row_off, int Tmax, int total_items, int* t_out, int* r_out) {
    const double us_per_step = 2.4, us_launch = 15.0, us_chain = 40.0, cu_share = 0.75;
    const double us_per_row = total_items > 0 ? 0.0007 + 5.15e-7 * (double)total_items : 0.011;
    const int64_t R = row_off[Tmax];
    double best = us_launch + us_per_step * Tmax + us_chain + us_per_row * (double)R;
    const double base = best;
    int at = 0, ar = 0;
    for (int t1 = 1; t1 < Tmax; ++t1) {
        const int64_t r1 = row_off[t1] / 64 * 64;
        if (r1 < 2048 || R - r1 < 2048) continue;
        const double g1 = us_launch + us_per_step * t1;
        const double g2 = us_launch + us_per_step * (Tmax - t1);
        const double c1 = us_chain + us_per_row * (double)r1 / cu_share;
        const double c2 = us_chain + us_per_row * (double)(R - r1);
        const double cost = g1 + (g2 > c1 ? g2 : c1) + c2;
        if (cost < best) {
            best = cost;
            at = t1;
            ar = (int)r1;
        }
    }
    if (!(best < 0.95 * base)) at = ar = 0;
    *t_out = at;
    *r_out = ar;
}

// a duplicate target inside a row counts once (scatter_ of 1.0, nb:510-513): all but the first become -1
void dedupe_row(int32_t* t, int K, std::vector<std::pair<int32_t, int32_t>>& tmp) {
    if (K <= 16) {
        for (int j = 1; j < K; ++j) {
            if (t[j] < 0) continue;
            for (int i = 0; i < j; ++i)
                if (t[i] == t[j]) {
                    t[j] = -1;
                    break;
                }
        }
        return;
    }
    tmp.clear();
    for (int j = 0; j < K; ++j)
        if (t[j] >= 0) tmp.emplace_back(t[j], j);
    std::sort(tmp.begin(), tmp.end());
    for (size_t i = 1; i < tmp.size(); ++i)
        if (tmp[i].first == tmp[i - 1].first) t[tmp[i].second] = -1;
}

template <class T>
long long put(char* dst, long long off, const std::vector<T>& v) {
    if (dst && !v.empty()) std::memcpy(dst + off, v.data(), v.size() * sizeof(T));
    return off + (long long)((v.size() * sizeof(T) + 255) / 256 * 256);
}

// the packed layout: every array at a multiple of 256 bytes, in batch.py's order
long long layout(const Plan& p, char* dst, long long* off) {
    long long o = 0;
    int i = 0;
#define SV_PUT(v_) off[i++] = o; o = put(dst, o, p.v_);
    SV_PUT(x) SV_PUT(tgt) SV_PUT(w_nll) SV_PUT(w_kl) SV_PUT(row_off) SV_PUT(nact) SV_PUT(last_row) SV_PUT(perm)
    SV_PUT(seg_off) SV_PUT(seg_item) SV_PUT(row_key) SV_PUT(seen) SV_PUT(seen_off)
    if (p.info.total_items > 0) { SV_PUT(occ_off) SV_PUT(occ_row) SV_PUT(occ_items) }
    else { off[i++] = -1; off[i++] = -1; off[i++] = -1; }
#undef SV_PUT
    return o > 256 ? o : 256;
}

}  // namespace

extern "C" void* svae_plan_create(const int* items, const long long* user_off, const long long* idx,
                                  const long long* keys, int B, int next_k, int total_items, svae_plan_info* info) {
    if (!items || !user_off || !idx || !info || B <= 0 || next_k <= 0 || total_items < 0) {
        svae_fail(-1, "svae_plan_create: bad argument (B=%d next_k=%d total_items=%d)", B, next_k, total_items);
        return nullptr;
    }
    Plan* pp = new Plan();
    Plan& p = *pp;
    std::vector<int64_t> lens((size_t)B);
    for (int u = 0; u < B; ++u) {
        lens[(size_t)u] = user_off[idx[u] + 1] - user_off[idx[u]] - 1;
        if (lens[(size_t)u] < 1) {
            svae_fail(-2, "every user needs at least 2 items (the reference crashes on T=0, nb:803)");
            delete pp;
            return nullptr;
        }
    }
    p.order.resize((size_t)B);
    std::iota(p.order.begin(), p.order.end(), (int64_t)0);
    std::stable_sort(p.order.begin(), p.order.end(),
                     [&](int64_t a, int64_t b) { return lens[(size_t)a] > lens[(size_t)b]; });
    p.lengths.resize((size_t)B);
    for (int u = 0; u < B; ++u) p.lengths[(size_t)u] = lens[(size_t)p.order[(size_t)u]];
    const int64_t* T = p.lengths.data();
    const int Tmax = (int)T[0];
    const int K = next_k < Tmax ? next_k : Tmax;

    // nact[t] = users still running at timestep t; rows of timestep t start at row_off[t]
    p.nact.assign((size_t)Tmax, 0);
    {
        std::vector<int32_t> ends((size_t)Tmax + 1, 0);
        for (int u = 0; u < B; ++u) ++ends[(size_t)T[u]];
        int32_t run = 0;
        for (int t = Tmax; t >= 1; --t) {
            run += ends[(size_t)t];
            p.nact[(size_t)t - 1] = run;
        }
    }
    p.row_off.assign((size_t)Tmax + 1, 0);
    for (int t = 0; t < Tmax; ++t) p.row_off[(size_t)t + 1] = p.row_off[(size_t)t] + p.nact[(size_t)t];
    const int64_t R = p.row_off[(size_t)Tmax];
    if (R * (int64_t)K > 0x7fffffffll || (int64_t)total_items > 0x3fffffffll) {
        svae_fail(-3, "svae_plan_create: batch too large (R=%lld K=%d)", (long long)R, K);
        delete pp;
        return nullptr;
    }

    p.x.resize((size_t)R);
    p.seen.resize((size_t)R);
    p.tgt.resize((size_t)(R * K));
    p.w_nll.resize((size_t)R);
    p.w_kl.resize((size_t)R);
    p.row_key.resize((size_t)R);
    p.seen_off.assign((size_t)B + 1, 0);
    p.last_row.resize((size_t)B);
    std::vector<std::pair<int32_t, int32_t>> tmp;
    int32_t max_item = 0;
    for (int u = 0; u < B; ++u) {
        const int64_t Tu = T[u];
        const int64_t src = idx[p.order[(size_t)u]];
        const int* it = items + user_off[src];
        const int64_t key = (keys ? keys[p.order[(size_t)u]] : p.order[(size_t)u]) << 20;
        const int32_t so = p.seen_off[(size_t)u];
        p.seen_off[(size_t)u + 1] = so + (int32_t)Tu;
        p.last_row[(size_t)u] = p.row_off[(size_t)Tu - 1] + u;
        const float wkl = (float)(1.0 / (double)((int64_t)B * Tu));
        double n0 = 1.0;
        for (int64_t t = 0; t < Tu; ++t) {
            const int64_t r = (int64_t)p.row_off[(size_t)t] + u;
            const int32_t xi = it[t];
            if (xi < 0 || (total_items > 0 && xi >= total_items)) {
                svae_fail(-4, "svae_plan_create: item id %d out of range [0, %d)", (int)xi, total_items);
                delete pp;
                return nullptr;
            }
            max_item = xi > max_item ? xi : max_item;
            p.x[(size_t)r] = xi;
            p.seen[(size_t)(so + t)] = xi;
            int32_t* tr = p.tgt.data() + r * K;
            for (int j = 0; j < K; ++j) tr[j] = t + j < Tu ? it[t + 1 + j] : -1;   // item t+1+j, if it exists
            dedupe_row(tr, K, tmp);
            if (t == 0) {                     // first timestep's target count (nb:803)
                int c = 0;
                for (int j = 0; j < K; ++j) c += tr[j] >= 0;
                n0 = (double)c;
            }
            p.w_nll[(size_t)r] = (float)(1.0 / ((double)B * n0));
            p.w_kl[(size_t)r] = wkl;
            p.row_key[(size_t)r] = key + t;
        }
        // (the last target of the user is never an input: range-check it as well)
        const int32_t li = it[Tu];
        if (li < 0 || (total_items > 0 && li >= total_items)) {
            svae_fail(-4, "svae_plan_create: item id %d out of range [0, %d)", (int)li, total_items);
            delete pp;
            return nullptr;
        }
    }

    // rows grouped by input item (item ascending, rows ascending inside an item), the groups ordered by length,
    // longest first (ties: item ascending): the embedding scatter-add gives the long ones a workgroup each
    {
        std::vector<uint32_t> key((size_t)R), val((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            key[(size_t)r] = (uint32_t)p.x[(size_t)r];
            val[(size_t)r] = (uint32_t)r;
        }
        radix_sort_pairs(key, val, (uint32_t)max_item);
        std::vector<int64_t> starts, slen;
        for (int64_t i = 0; i < R; ++i)
            if (i == 0 || key[(size_t)i] != key[(size_t)i - 1]) starts.push_back(i);
        const int64_t ns = (int64_t)starts.size();
        slen.resize((size_t)ns);
        int64_t mx = 0;
        for (int64_t s = 0; s < ns; ++s) {
            slen[(size_t)s] = (s + 1 < ns ? starts[(size_t)s + 1] : R) - starts[(size_t)s];
            mx = slen[(size_t)s] > mx ? slen[(size_t)s] : mx;
        }
        std::vector<int32_t> by_len;
        order_by_count_desc(slen.data(), ns, mx, by_len);
        p.seg_off.assign((size_t)ns + 1, 0);
        p.seg_item.resize((size_t)ns);
        p.perm.resize((size_t)R);
        int nlong = 0;
        int32_t o = 0;
        for (int64_t s = 0; s < ns; ++s) {
            const int64_t g = by_len[(size_t)s], st = starts[(size_t)g], n = slen[(size_t)g];
            p.seg_item[(size_t)s] = (int32_t)key[(size_t)st];
            for (int64_t i = 0; i < n; ++i) p.perm[(size_t)(o + i)] = (int32_t)val[(size_t)(st + i)];
            o += (int32_t)n;
            p.seg_off[(size_t)s + 1] = o;
            nlong += n > EMBED_LONG_ROWS;
        }
        p.info.nseg = (int)ns;
        p.info.nseg_long = nlong;
    }

    // target occurrences grouped by item (rows ascending inside an item; ties: target column ascending), and the
    // items by occurrence count, busiest first (ties: item ascending)
    p.info.nocc = 0;
    p.info.nocc_long = 0;
    if (total_items > 0) {
        // one counting sort over the catalogue (occ_off is needed anyway): count, prefix, scatter in ascending
        // (row, column) order
        const int64_t nq = R * K;
        const int32_t* tg = p.tgt.data();
        std::vector<int32_t> cur((size_t)total_items + 1, 0);
        int64_t no = 0;
        for (int64_t q = 0; q < nq; ++q) {
            const int32_t t = tg[q];
            if (t < 0) continue;
            if (t >= total_items) {
                svae_fail(-4, "svae_plan_create: item id %d out of range [0, %d)", (int)t, total_items);
                delete pp;
                return nullptr;
            }
            ++cur[(size_t)t + 1];
            ++no;
        }
        std::vector<int64_t> cnt((size_t)total_items);
        int64_t mx = 0;
        int nlong = 0;
        for (int i = 0; i < total_items; ++i) {
            const int64_t c = cur[(size_t)i + 1];
            cnt[(size_t)i] = c;
            mx = c > mx ? c : mx;
            nlong += c > OCC_LONG;
            cur[(size_t)i + 1] += cur[(size_t)i];
        }
        p.occ_off = cur;
        p.occ_row.resize((size_t)no);
        for (int64_t r = 0; r < R; ++r)
            for (int j = 0; j < K; ++j) {
                const int32_t t = tg[r * K + j];
                if (t >= 0) p.occ_row[(size_t)cur[(size_t)t]++] = (int32_t)r;
            }
        order_by_count_desc(cnt.data(), total_items, mx, p.occ_items);
        p.info.nocc = (int)no;
        p.info.nocc_long = nlong;
    }

    p.info.B = B;
    p.info.R = (int)R;
    p.info.Tmax = Tmax;
    p.info.K = K;
    p.info.total_items = total_items;
    choose_time_split(p.row_off.data(), Tmax, total_items, &p.info.split_t, &p.info.split_r);
    p.info.total_bytes = layout(p, nullptr, p.info.off);
    *info = p.info;
    return pp;
}

extern "C" int svae_plan_pack(void* plan, void* dst, long long dst_bytes, long long* order, long long* lengths,
                              int* row_off) {
    SV_CHECK_ARG(plan && dst, "svae_plan_pack: null argument");
    const Plan& p = *static_cast<const Plan*>(plan);
    SV_CHECK_ARG(dst_bytes >= p.info.total_bytes, "svae_plan_pack: buffer of %lld bytes, plan needs %lld", dst_bytes,
                 p.info.total_bytes);
    long long off[16];
    layout(p, static_cast<char*>(dst), off);
    if (order) std::memcpy(order, p.order.data(), p.order.size() * sizeof(int64_t));
    if (lengths) std::memcpy(lengths, p.lengths.data(), p.lengths.size() * sizeof(int64_t));
    if (row_off) std::memcpy(row_off, p.row_off.data(), p.row_off.size() * sizeof(int32_t));
    return 0;
}

extern "C" void svae_plan_destroy(void* plan) { delete static_cast<Plan*>(plan); }
