// torch.ops.hopwise_amd.* -- the C ABI of include/hopwise_amd.h registered with PyTorch's dispatcher (SURVEY.md §8(b)
// "what the native library must export": a TORCH_LIBRARY namespace over the extern "C" launchers).
//
// Every op is a thin call into the hw_* launchers of libhopwise_amd.so: it allocates the outputs with torch, takes the
// current HIP stream, and raises on a non-zero status.  No arithmetic lives here.  The ops mirror how the reference
// uses its scorers (model/knowledge_graph_embedding_recommender/rotate.py:61-69 forward on gathered rows, :98-131
// calculate_loss, :161-220 full sorts; sampler/sampler.py:140-183; knowledge_aware_recommender/pgpr.py:440-552).
// Autograd for the differentiable ops is registered from Python (hopwise_amd/torch_ops.py, torch.library).
//
// `pair_loss` / `pair_loss_backward` are the whole calculate_loss of the pairwise gather models as ONE op each way
// (both sides' scorers -> loss kernel -> dLoss/dscore; both sides' gradient launches): an eager trainer step crosses
// from Python into C++ twice instead of ~20 times.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>
#include <vector>

#include "../../include/hopwise_amd.h"

namespace {

using at::Tensor;
using TensorList = at::TensorList;

void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void check(int rc, const char* what) {
    TORCH_CHECK(rc == HW_OK, "hopwise_amd: ", what, " failed with status ", rc,
                rc == HW_ERR_INVALID_ARG ? " (invalid argument)" : rc == HW_ERR_UNSUPPORTED ? " (unsupported)"
                : rc == HW_ERR_LAUNCH ? " (launch error)" : rc == HW_ERR_WORKSPACE ? " (workspace too small)" : "");
}

const Tensor& dev_f32(const Tensor& t, const char* what) {
    TORCH_CHECK(t.is_cuda(), "hopwise_amd: ", what, ": expected a HIP tensor; there is no CPU path");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "hopwise_amd: ", what, ": expected float32");
    return t;
}
const Tensor& dev_i64(const Tensor& t, const char* what) {
    TORCH_CHECK(t.is_cuda(), "hopwise_amd: ", what, ": expected a HIP tensor; there is no CPU path");
    TORCH_CHECK(t.scalar_type() == at::kLong, "hopwise_amd: ", what, ": expected int64");
    return t;
}

// per-device out-of-range-id counter shared by the ops of this library (read with oob_count)
Tensor& oob_counter(const Tensor& like) {
    static std::vector<Tensor> per_dev(64);
    const int d = like.get_device();
    TORCH_CHECK(d >= 0 && d < 64, "device index");
    if (!per_dev[d].defined()) per_dev[d] = at::zeros({1}, like.options().dtype(at::kInt));
    return per_dev[d];
}

hw_kge_desc make_desc(int64_t model, TensorList head, TensorList rel, TensorList tail, double margin, double eps,
                      int64_t tail_rows) {
    hw_kge_desc d{};
    d.model = (int32_t)model;
    TORCH_CHECK(!head.empty() && !rel.empty() && !tail.empty() && head.size() <= HW_KGE_MAX_TABLES &&
                    rel.size() <= HW_KGE_MAX_TABLES && tail.size() <= HW_KGE_MAX_TABLES,
                "hopwise_amd: 1..3 tables per slot");
    const int64_t dim = head[0].size(1);
    auto fill = [&](TensorList tabs, const float** dst, const char* what, bool rel_slot) {
        const int64_t rows = tabs[0].size(0);
        for (size_t k = 0; k < tabs.size(); ++k) {
            dev_f32(tabs[k], what);
            const int64_t want = (model == HW_RESCAL && rel_slot) ? dim * dim : dim;
            TORCH_CHECK(tabs[k].dim() == 2 && tabs[k].size(1) == want && tabs[k].size(0) == rows && tabs[k].is_contiguous(),
                        "hopwise_amd: ", what, " table ", k, ": need contiguous [rows, ", want, "]");
            dst[k] = tabs[k].data_ptr<float>();
        }
        return rows;
    };
    d.dim = (int32_t)dim;
    d.n_head = fill(head, d.head, "head", false);
    d.n_rel = fill(rel, d.rel, "rel", true);
    d.n_tail = fill(tail, d.tail, "tail", false);
    if (tail_rows > 0) {
        TORCH_CHECK(tail_rows <= d.n_tail, "hopwise_amd: tail_rows out of range");
        d.n_tail = tail_rows;
    }
    d.margin = (float)margin;
    d.eps = (float)eps;
    return d;
}

const int64_t* opt_i64(const c10::optional<Tensor>& t, const char* what) {
    return t.has_value() && t->defined() ? dev_i64(*t, what).data_ptr<int64_t>() : nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------
Tensor kge_score_grouped(int64_t model, TensorList head, TensorList rel, TensorList tail, const Tensor& h_idx,
                         const c10::optional<Tensor>& r_idx, int64_t r_fixed, const Tensor& t_idx, double margin, double eps,
                         bool j_major_out, int64_t tail_rows) {
    const hw_kge_desc d = make_desc(model, head, rel, tail, margin, eps, tail_rows);
    const Tensor h = dev_i64(h_idx, "h_idx").contiguous();
    dev_i64(t_idx, "t_idx");
    TORCH_CHECK(t_idx.dim() == 2 && t_idx.size(0) == h.size(0), "hopwise_amd: t_idx must be [n_groups, n_tail_per_group]");
    c10::optional<Tensor> r = r_idx;
    if (r.has_value() && r->defined()) r = r->contiguous();
    const int64_t B = t_idx.size(0), G = t_idx.size(1);
    Tensor out = j_major_out ? at::empty({G, B}, head[0].options()).t() : at::empty({B, G}, head[0].options());
    check(hw_kge_score_grouped(&d, h.data_ptr<int64_t>(), opt_i64(r, "r_idx"), r_fixed, t_idx.data_ptr<int64_t>(),
                               t_idx.stride(0), t_idx.stride(1), B, G, out.data_ptr<float>(), out.stride(0), out.stride(1),
                               oob_counter(h).data_ptr<int32_t>(), cur_stream(h)),
          "hw_kge_score_grouped");
    return out;
}

// dense gradients of the tables, in head + rel + tail order; an entry whose `need` is false comes back empty ([0])
std::vector<Tensor> kge_score_grouped_backward(int64_t model, TensorList head, TensorList rel, TensorList tail,
                                               const Tensor& h_idx, const c10::optional<Tensor>& r_idx, int64_t r_fixed,
                                               const Tensor& t_idx, double margin, double eps, int64_t tail_rows,
                                               const Tensor& grad, at::IntArrayRef need) {
    const hw_kge_desc d = make_desc(model, head, rel, tail, margin, eps, tail_rows);
    const size_t n = head.size() + rel.size() + tail.size();
    TORCH_CHECK(need.size() == n, "hopwise_amd: one `need` flag per table");
    const Tensor h = dev_i64(h_idx, "h_idx").contiguous();
    c10::optional<Tensor> r = r_idx;
    if (r.has_value() && r->defined()) r = r->contiguous();
    dev_f32(grad, "grad");
    const int64_t B = t_idx.size(0), G = t_idx.size(1);
    TORCH_CHECK(grad.dim() == 2 && grad.size(0) == B && grad.size(1) == G, "hopwise_amd: grad must be [n_groups, n_tail]");
    hw_kge_grads gs{};
    std::vector<Tensor> outs;
    std::vector<std::pair<const void*, Tensor>> shared;  // a parameter used in two slots (entity table as head and tail)
    size_t p = 0;
    auto slot = [&](TensorList tabs, float** dst) {
        for (size_t k = 0; k < tabs.size(); ++k, ++p) {
            if (!need[p]) {
                outs.push_back(at::empty({0}, tabs[k].options()));
                continue;
            }
            Tensor buf;
            for (auto& s : shared)
                if (s.first == tabs[k].data_ptr()) buf = s.second;
            const bool fresh = !buf.defined();
            if (fresh) {
                buf = at::zeros_like(tabs[k]);
                shared.emplace_back(tabs[k].data_ptr(), buf);
            }
            dst[k] = buf.data_ptr<float>();
            outs.push_back(fresh ? buf : at::empty({0}, tabs[k].options()));  // the shared buffer is returned once
        }
    };
    slot(head, gs.head);
    slot(rel, gs.rel);
    slot(tail, gs.tail);
    const bool fixed = !(r.has_value() && r->defined());
    const int64_t nws = hw_kge_grad_grouped_workspace_floats(&d, B, G, fixed ? 1 : 0);
    Tensor ws;
    if (nws > 0) ws = at::empty({nws}, head[0].options());
    check(hw_kge_grad_grouped(&d, &gs, h.data_ptr<int64_t>(), opt_i64(r, "r_idx"), r_fixed, t_idx.data_ptr<int64_t>(),
                              t_idx.stride(0), t_idx.stride(1), B, G, grad.data_ptr<float>(), grad.stride(0), grad.stride(1),
                              nws > 0 ? ws.data_ptr<float>() : nullptr, cur_stream(h)),
          "hw_kge_grad_grouped");
    return outs;
}

Tensor kge_full_sort(int64_t model, TensorList head, TensorList rel, TensorList tail, const Tensor& h_idx,
                     const c10::optional<Tensor>& r_idx, int64_t r_fixed, int64_t lo, int64_t hi, double margin) {
    const hw_kge_desc d = make_desc(model, head, rel, tail, margin, 0.0, 0);
    const Tensor h = dev_i64(h_idx, "h_idx").contiguous();
    c10::optional<Tensor> r = r_idx;
    if (r.has_value() && r->defined()) r = r->contiguous();
    TORCH_CHECK(0 <= lo && lo < hi, "hopwise_amd: empty tail range");
    Tensor out = at::empty({h.size(0), hi - lo}, head[0].options());
    check(hw_kge_full_sort(&d, h.data_ptr<int64_t>(), opt_i64(r, "r_idx"), r_fixed, h.size(0), lo, hi, out.data_ptr<float>(),
                           hi - lo, oob_counter(h).data_ptr<int32_t>(), cur_stream(h)),
          "hw_kge_full_sort");
    return out;
}

std::tuple<Tensor, Tensor> kge_full_sort_topk(int64_t model, TensorList head, TensorList rel, TensorList tail,
                                              const Tensor& h_idx, const c10::optional<Tensor>& r_idx, int64_t r_fixed,
                                              int64_t lo, int64_t hi, int64_t k, const c10::optional<Tensor>& hist_rows,
                                              const c10::optional<Tensor>& hist_cols, bool mask_first_col, double margin) {
    const hw_kge_desc d = make_desc(model, head, rel, tail, margin, 0.0, 0);
    const Tensor h = dev_i64(h_idx, "h_idx").contiguous();
    c10::optional<Tensor> r = r_idx, hr = hist_rows, hc = hist_cols;
    if (r.has_value() && r->defined()) r = r->contiguous();
    int64_t nh = 0;
    if (hr.has_value() && hr->defined()) {
        TORCH_CHECK(hc.has_value() && hc->defined() && hc->numel() == hr->numel(), "hopwise_amd: history rows / cols differ");
        hr = hr->contiguous();
        hc = hc->contiguous();
        nh = hr->numel();
    }
    const int64_t B = h.size(0);
    const int64_t nbytes = hw_kge_full_sort_topk_workspace_bytes(B, hi - lo, (int32_t)k);
    Tensor ws = at::empty({nbytes}, head[0].options().dtype(at::kByte));
    Tensor val = at::empty({B, k}, head[0].options());
    Tensor idx = at::empty({B, k}, h.options());
    check(hw_kge_full_sort_topk(&d, h.data_ptr<int64_t>(), opt_i64(r, "r_idx"), r_fixed, B, lo, hi,
                                nh ? hr->data_ptr<int64_t>() : nullptr, nh ? hc->data_ptr<int64_t>() : nullptr, nh,
                                mask_first_col ? 1 : 0, (int32_t)k, ws.data_ptr(), nbytes, val.data_ptr<float>(),
                                idx.data_ptr<int64_t>(), cur_stream(h)),
          "hw_kge_full_sort_topk");
    return {val, idx};
}

// value_ids[j * len(keys) + i] = j-th negative of keys[i] (sampler.py:148-152); seed / offset are the Philox key / counter
Tensor kg_neg_sample(const Tensor& used_ptr, const Tensor& used_idx, const Tensor& keys, int64_t num, int64_t value_num,
                     int64_t seed, int64_t offset, const c10::optional<Tensor>& alias_prob,
                     const c10::optional<Tensor>& alias_keys, const c10::optional<Tensor>& alias_alias) {
    const Tensor up = dev_i64(used_ptr, "used_ptr").contiguous(), ui = dev_i64(used_idx, "used_idx").contiguous();
    const Tensor k = dev_i64(keys, "keys").contiguous();
    const int64_t n = k.numel();
    Tensor out = at::empty({num * n}, k.options());
    const float* ap = nullptr;
    const int64_t *ak = nullptr, *aa = nullptr;
    int64_t na = 0;
    if (alias_prob.has_value() && alias_prob->defined()) {
        ap = dev_f32(*alias_prob, "alias_prob").data_ptr<float>();
        ak = dev_i64(*alias_keys, "alias_keys").data_ptr<int64_t>();
        aa = dev_i64(*alias_alias, "alias_alias").data_ptr<int64_t>();
        na = alias_prob->numel();
    }
    check(hw_neg_sample(up.data_ptr<int64_t>(), ui.data_ptr<int64_t>(), up.numel() - 1, k.data_ptr<int64_t>(), n,
                        (int32_t)num, value_num, ap, ak, aa, na, (uint64_t)seed, (uint64_t)offset, out.data_ptr<int64_t>(),
                        1, n, oob_counter(k).data_ptr<int32_t>(), cur_stream(k)),
          "hw_neg_sample");
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// calculate_loss of the pairwise gather models, one op.  `sides` is a flat int list, per side:
//   [n_head, head positions..., n_rel, rel positions..., n_tail, tail positions..., r_fixed, has_r_idx, tail_rows]
// with positions indexing `params`; `idx` holds per side h_idx, (r_idx when has_r_idx), t_idx [B, G].
// kind 0: sum over sides of BCEWithLogits(mean) with the B positives first, positives weighted pos_weight[side];
// kind 1 / 2: MarginRankingLoss / its sigmoid form over the concatenated sides ((positive, negative) pairs only).
// Returns (loss scalar, dLoss/dscore buffer for pair_loss_backward).
struct Side {
    std::vector<int64_t> head, rel, tail;
    int64_t r_fixed, has_r, tail_rows;
    Tensor h, r, t;
};

std::vector<Side> parse_sides(at::IntArrayRef sides, TensorList idx) {
    std::vector<Side> out;
    size_t i = 0, j = 0;
    while (i < sides.size()) {
        Side s;
        for (auto* v : {&s.head, &s.rel, &s.tail}) {
            const int64_t n = sides[i++];
            TORCH_CHECK(n >= 1 && n <= HW_KGE_MAX_TABLES && i + n <= sides.size(), "hopwise_amd: malformed side layout");
            for (int64_t k = 0; k < n; ++k) v->push_back(sides[i++]);
        }
        TORCH_CHECK(i + 3 <= sides.size(), "hopwise_amd: malformed side layout");
        s.r_fixed = sides[i++], s.has_r = sides[i++], s.tail_rows = sides[i++];
        TORCH_CHECK(j + 2 + (s.has_r ? 1 : 0) <= idx.size(), "hopwise_amd: too few index tensors");
        s.h = dev_i64(idx[j++], "h_idx").contiguous();
        if (s.has_r) s.r = dev_i64(idx[j++], "r_idx").contiguous();
        s.t = dev_i64(idx[j++], "t_idx");
        TORCH_CHECK(s.t.dim() == 2 && s.t.size(0) == s.h.size(0), "hopwise_amd: t_idx must be [n_groups, n_tail_per_group]");
        out.push_back(std::move(s));
    }
    return out;
}

hw_kge_desc side_desc(int64_t model, const Side& s, TensorList params, double margin, double eps) {
    std::vector<Tensor> h, r, t;
    for (auto p : s.head) h.push_back(params[p]);
    for (auto p : s.rel) r.push_back(params[p]);
    for (auto p : s.tail) t.push_back(params[p]);
    return make_desc(model, h, r, t, margin, eps, s.tail_rows);
}

std::tuple<Tensor, Tensor> pair_loss(int64_t model, int64_t kind, double margin, double eps, double scale,
                                     TensorList params, at::IntArrayRef sides, TensorList idx,
                                     at::ArrayRef<double> pos_weight, bool need_grad) {
    auto S = parse_sides(sides, idx);
    TORCH_CHECK(!S.empty() && pos_weight.size() == S.size(), "hopwise_amd: one pos_weight per side");
    const auto fopt = dev_f32(params[0], "params").options();
    void* st = cur_stream(params[0]);
    int32_t* oob = oob_counter(params[0]).data_ptr<int32_t>();
    Tensor ws = at::empty({hw_loss_workspace_floats()}, fopt);
    Tensor loss, g;
    if (kind == 0) {
        int64_t total = 0;
        for (auto& s : S) total += s.t.size(0) * s.t.size(1);
        Tensor buf = at::empty({total}, fopt);
        Tensor losses = at::empty({(int64_t)S.size()}, fopt);
        if (need_grad) g = at::empty({total}, fopt);
        int64_t off = 0;
        for (size_t k = 0; k < S.size(); ++k) {
            auto& s = S[k];
            const hw_kge_desc d = side_desc(model, s, params, margin, eps);
            const int64_t B = s.t.size(0), G = s.t.size(1);
            float* b = buf.data_ptr<float>() + off;  // [G, B] j-major: the B positives first
            check(hw_kge_score_grouped(&d, s.h.data_ptr<int64_t>(), s.has_r ? s.r.data_ptr<int64_t>() : nullptr, s.r_fixed,
                                       s.t.data_ptr<int64_t>(), s.t.stride(0), s.t.stride(1), B, G, b, 1, B, oob, st),
                  "hw_kge_score_grouped");
            check(hw_loss_bce_logits_weighted(b, G * B, B, (float)pos_weight[k], losses.data_ptr<float>() + k,
                                              need_grad ? g.data_ptr<float>() + off : nullptr, ws.data_ptr<float>(), st),
                  "hw_loss_bce_logits_weighted");
            off += B * G;
        }
        loss = S.size() > 1 ? losses.sum() : losses[0];
    } else {
        int64_t n_tot = 0;
        for (auto& s : S) {
            TORCH_CHECK(s.t.size(1) == 2, "hopwise_amd: margin losses take (positive, negative) pairs");
            n_tot += s.t.size(0);
        }
        Tensor buf = at::empty({2, n_tot}, fopt);
        int64_t off = 0;
        for (auto& s : S) {
            const hw_kge_desc d = side_desc(model, s, params, margin, eps);
            const int64_t B = s.t.size(0);
            check(hw_kge_score_grouped(&d, s.h.data_ptr<int64_t>(), s.has_r ? s.r.data_ptr<int64_t>() : nullptr, s.r_fixed,
                                       s.t.data_ptr<int64_t>(), s.t.stride(0), s.t.stride(1), B, 2,
                                       buf.data_ptr<float>() + off, 1, n_tot, oob, st),
                  "hw_kge_score_grouped");
            off += B;
        }
        if (need_grad) g = at::empty({2, n_tot}, fopt);
        Tensor losses = at::empty({1}, fopt);
        float* g0 = need_grad ? g.data_ptr<float>() : nullptr;
        float* g1 = need_grad ? g.data_ptr<float>() + n_tot : nullptr;
        const float* p = buf.data_ptr<float>();
        if (kind == 2)
            check(hw_loss_margin_ranking_sigmoid(p, p + n_tot, n_tot, (float)margin, losses.data_ptr<float>(), g0, g1,
                                                 ws.data_ptr<float>(), st), "hw_loss_margin_ranking_sigmoid");
        else
            check(hw_loss_margin_ranking(p, p + n_tot, n_tot, (float)margin, losses.data_ptr<float>(), g0, g1,
                                         ws.data_ptr<float>(), st), "hw_loss_margin_ranking");
        loss = losses[0];
    }
    if (scale != 1.0) loss = loss * scale;
    if (!g.defined()) g = at::empty({0}, fopt);
    return {loss, g};
}

// Gradient launches of pair_loss.  `sinks[p]` (optional, per parameter): a persistent gradient buffer to accumulate into
// (optim.FusedAdam's p.grad); parameters without one get a zero-filled dense gradient, returned at their position
// (empty [0] tensors elsewhere).
std::vector<Tensor> pair_loss_backward(int64_t model, int64_t kind, double margin, double eps, double scale,
                                       TensorList params, at::IntArrayRef sides, TensorList idx, const Tensor& dscore,
                                       const Tensor& g_up, at::IntArrayRef needs,
                                       const c10::List<c10::optional<Tensor>>& sinks) {
    auto S = parse_sides(sides, idx);
    TORCH_CHECK(needs.size() == params.size() && sinks.size() == params.size(), "hopwise_amd: one flag / sink per parameter");
    const auto fopt = params[0].options();
    void* st = cur_stream(params[0]);
    Tensor gall = dscore * (scale != 1.0 ? g_up * scale : g_up);  // every side's dLoss/dscore x upstream, one launch
    std::vector<Tensor> bufs(params.size()), outs(params.size());
    for (size_t p = 0; p < params.size(); ++p) outs[p] = at::empty({0}, fopt);
    auto buffer = [&](int64_t p) -> float* {
        if (!needs[p]) return nullptr;
        if (!bufs[p].defined()) {
            for (size_t q = 0; q < params.size(); ++q)  // the same storage registered under two positions
                if (bufs[q].defined() && params[q].data_ptr() == params[p].data_ptr()) bufs[p] = bufs[q];
        }
        if (!bufs[p].defined()) {
            c10::optional<Tensor> sk = sinks.get(p);
            if (sk.has_value() && sk->defined()) {
                bufs[p] = *sk;
            } else {
                bufs[p] = at::zeros_like(params[p]);
                outs[p] = bufs[p];
            }
        }
        return bufs[p].data_ptr<float>();
    };
    int64_t off = 0, n_tot = 0;
    if (kind != 0)
        for (auto& s : S) n_tot += s.t.size(0);
    for (auto& s : S) {
        const hw_kge_desc d = side_desc(model, s, params, margin, eps);
        hw_kge_grads gs{};
        for (size_t k = 0; k < s.head.size(); ++k) gs.head[k] = buffer(s.head[k]);
        for (size_t k = 0; k < s.rel.size(); ++k) gs.rel[k] = buffer(s.rel[k]);
        for (size_t k = 0; k < s.tail.size(); ++k) gs.tail[k] = buffer(s.tail[k]);
        const int64_t B = s.t.size(0), G = s.t.size(1);
        const float* gp = gall.data_ptr<float>() + off;
        const int64_t gsj = kind == 0 ? B : n_tot;
        off += kind == 0 ? B * G : B;
        const int64_t nws = hw_kge_grad_grouped_workspace_floats(&d, B, G, s.has_r ? 0 : 1);
        Tensor ws;
        if (nws > 0) ws = at::empty({nws}, fopt);
        check(hw_kge_grad_grouped(&d, &gs, s.h.data_ptr<int64_t>(), s.has_r ? s.r.data_ptr<int64_t>() : nullptr, s.r_fixed,
                                  s.t.data_ptr<int64_t>(), s.t.stride(0), s.t.stride(1), B, G, gp, 1, gsj,
                                  nws > 0 ? ws.data_ptr<float>() : nullptr, st),
              "hw_kge_grad_grouped");
    }
    return outs;
}

// ---------------------------------------------------------------------------------------------------------------------
hw_pgpr_graph make_graph(const Tensor& ptr, const Tensor& rel, const Tensor& dst, at::IntArrayRef meta) {
    TORCH_CHECK(meta.size() == 5, "hopwise_amd: graph meta = [n_users, n_entities, n_items, ui_relation, max_degree]");
    TORCH_CHECK(ptr.is_cuda() && ptr.scalar_type() == at::kLong && rel.scalar_type() == at::kInt &&
                    dst.scalar_type() == at::kInt, "hopwise_amd: CSR = int64 ptr, int32 rel / dst on the device");
    hw_pgpr_graph g{};
    g.ptr = ptr.data_ptr<int64_t>();
    g.rel = rel.data_ptr<int32_t>();
    g.dst = dst.data_ptr<int32_t>();
    g.n_users = (int32_t)meta[0], g.n_entities = (int32_t)meta[1], g.n_items = (int32_t)meta[2];
    g.ui_relation = (int32_t)meta[3], g.max_degree = (int32_t)meta[4];
    return g;
}

// children of every beam row in row order, top-k order inside a row (pgpr.py:460-479): (paths_out, probs_out, n_out)
std::tuple<Tensor, Tensor, Tensor> pgpr_expand(const Tensor& g_ptr, const Tensor& g_rel, const Tensor& g_dst,
                                               at::IntArrayRef g_meta, const Tensor& paths, const Tensor& probs,
                                               const Tensor& acts, const Tensor& n_acts, const Tensor& n_rows,
                                               const Tensor& topk_idx, const Tensor& topk_val, int64_t out_capacity) {
    const hw_pgpr_graph g = make_graph(g_ptr, g_rel, g_dst, g_meta);
    const int64_t max_rows = paths.size(0), act_dim = acts.size(1), k = topk_idx.size(1);
    TORCH_CHECK(paths.scalar_type() == at::kInt && paths.size(1) == HW_PGPR_ROW_INTS && paths.is_contiguous() &&
                    acts.scalar_type() == at::kInt && acts.is_contiguous() && topk_idx.scalar_type() == at::kInt &&
                    topk_idx.is_contiguous() && topk_val.is_contiguous() && probs.is_contiguous(),
                "hopwise_amd: pgpr_expand operand layout");
    auto iopt = paths.options();
    Tensor offsets = at::empty({max_rows}, iopt), sums = at::empty({(max_rows + 1023) / 1024 + 1}, iopt);
    Tensor pout = at::empty({out_capacity, HW_PGPR_ROW_INTS}, iopt), qout = at::empty({out_capacity, 3}, probs.options());
    Tensor nout = at::zeros({1}, iopt);
    check(hw_pgpr_expand(&g, paths.data_ptr<int32_t>(), probs.data_ptr<float>(), acts.data_ptr<int32_t>(), (int32_t)act_dim,
                         n_acts.data_ptr<int32_t>(), n_rows.data_ptr<int32_t>(), max_rows, topk_idx.data_ptr<int32_t>(),
                         topk_val.data_ptr<float>(), (int32_t)k, offsets.data_ptr<int32_t>(), sums.data_ptr<int32_t>(),
                         pout.data_ptr<int32_t>(), qout.data_ptr<float>(), nout.data_ptr<int32_t>(), out_capacity,
                         cur_stream(paths)),
          "hw_pgpr_expand");
    return {pout, qout, nout};
}

// collect_scores (pgpr.py:486-552): (results [n_batch, n_items], sel_item, sel_score, sel_leaf, sel_count, sel_row)
std::vector<Tensor> pgpr_collect(const Tensor& g_ptr, const Tensor& g_rel, const Tensor& g_dst, at::IntArrayRef g_meta,
                                 const Tensor& user64, const Tensor& entity64, const Tensor& relation64, const Tensor& users,
                                 const Tensor& positives_bits, const Tensor& leaf_paths, const Tensor& leaf_probs,
                                 const Tensor& n_leaf, int64_t n_nodes, int64_t kmax) {
    const hw_pgpr_graph g = make_graph(g_ptr, g_rel, g_dst, g_meta);
    TORCH_CHECK(user64.scalar_type() == at::kDouble && entity64.scalar_type() == at::kDouble &&
                    relation64.scalar_type() == at::kDouble && user64.is_contiguous() && entity64.is_contiguous() &&
                    relation64.is_contiguous(), "hopwise_amd: float64 contiguous tables");
    const Tensor u = dev_i64(users, "users").contiguous();
    const int64_t nb = u.numel(), ni = g.n_items, dim = user64.size(1);
    auto fopt = leaf_probs.options();
    auto iopt = leaf_paths.options();
    Tensor best = at::empty({nb * ni}, u.options()), score = at::empty({nb * ni}, user64.options());
    Tensor first = at::empty({nb}, iopt);
    Tensor results = at::empty({nb, ni}, fopt);
    Tensor sel_item = at::empty({nb, kmax}, iopt), sel_score = at::empty({nb, kmax}, user64.options());
    Tensor sel_leaf = at::empty({nb, kmax}, iopt), sel_count = at::empty({nb}, iopt), sel_row = at::empty({nb}, iopt);
    check(hw_pgpr_collect(&g, user64.data_ptr<double>(), entity64.data_ptr<double>(), relation64.data_ptr<double>(),
                          (int32_t)dim, u.data_ptr<int64_t>(), nb, (const uint32_t*)positives_bits.data_ptr(),
                          positives_bits.size(1), leaf_paths.data_ptr<int32_t>(), leaf_probs.data_ptr<float>(),
                          n_leaf.data_ptr<int32_t>(), leaf_paths.size(0), (int32_t)n_nodes, (int32_t)kmax,
                          (uint64_t*)best.data_ptr<int64_t>(), score.data_ptr<double>(), first.data_ptr<int32_t>(),
                          results.data_ptr<float>(), sel_item.data_ptr<int32_t>(), sel_score.data_ptr<double>(),
                          sel_leaf.data_ptr<int32_t>(), sel_count.data_ptr<int32_t>(), sel_row.data_ptr<int32_t>(),
                          cur_stream(u)),
          "hw_pgpr_collect");
    return {results, sel_item, sel_score, sel_leaf, sel_count, sel_row};
}

int64_t oob_count(const Tensor& like) { return oob_counter(like).item<int32_t>(); }

}  // namespace

TORCH_LIBRARY(hopwise_amd, m) {
    m.def("kge_score_grouped(int model, Tensor[] head, Tensor[] rel, Tensor[] tail, Tensor h_idx, Tensor? r_idx, int r_fixed, "
          "Tensor t_idx, float margin, float eps, bool j_major_out, int tail_rows) -> Tensor");
    m.def("kge_score_grouped_backward(int model, Tensor[] head, Tensor[] rel, Tensor[] tail, Tensor h_idx, Tensor? r_idx, "
          "int r_fixed, Tensor t_idx, float margin, float eps, int tail_rows, Tensor grad, int[] need) -> Tensor[]");
    m.def("kge_full_sort(int model, Tensor[] head, Tensor[] rel, Tensor[] tail, Tensor h_idx, Tensor? r_idx, int r_fixed, "
          "int lo, int hi, float margin) -> Tensor");
    m.def("kge_full_sort_topk(int model, Tensor[] head, Tensor[] rel, Tensor[] tail, Tensor h_idx, Tensor? r_idx, int r_fixed, "
          "int lo, int hi, int k, Tensor? hist_rows, Tensor? hist_cols, bool mask_first_col, float margin) -> (Tensor, Tensor)");
    m.def("kg_neg_sample(Tensor used_ptr, Tensor used_idx, Tensor keys, int num, int value_num, int seed, int offset, "
          "Tensor? alias_prob, Tensor? alias_keys, Tensor? alias_alias) -> Tensor");
    m.def("pair_loss(int model, int kind, float margin, float eps, float scale, Tensor[] params, int[] sides, Tensor[] idx, "
          "float[] pos_weight, bool need_grad) -> (Tensor, Tensor)");
    m.def("pair_loss_backward(int model, int kind, float margin, float eps, float scale, Tensor[] params, int[] sides, "
          "Tensor[] idx, Tensor dscore, Tensor g_up, int[] needs, Tensor?[] sinks) -> Tensor[]");
    m.def("pgpr_expand(Tensor g_ptr, Tensor g_rel, Tensor g_dst, int[] g_meta, Tensor paths, Tensor probs, Tensor acts, "
          "Tensor n_acts, Tensor n_rows, Tensor topk_idx, Tensor topk_val, int out_capacity) -> (Tensor, Tensor, Tensor)");
    m.def("pgpr_collect(Tensor g_ptr, Tensor g_rel, Tensor g_dst, int[] g_meta, Tensor user64, Tensor entity64, "
          "Tensor relation64, Tensor users, Tensor positives_bits, Tensor leaf_paths, Tensor leaf_probs, Tensor n_leaf, "
          "int n_nodes, int kmax) -> Tensor[]");
    m.def("oob_count(Tensor like) -> int");
}

TORCH_LIBRARY_IMPL(hopwise_amd, CUDA, m) {  // the dispatch key of HIP tensors on a ROCm build
    m.impl("kge_score_grouped", &kge_score_grouped);
    m.impl("kge_score_grouped_backward", &kge_score_grouped_backward);
    m.impl("kge_full_sort", &kge_full_sort);
    m.impl("kge_full_sort_topk", &kge_full_sort_topk);
    m.impl("kg_neg_sample", &kg_neg_sample);
    m.impl("pair_loss", &pair_loss);
    m.impl("pair_loss_backward", &pair_loss_backward);
    m.impl("pgpr_expand", &pgpr_expand);
    m.impl("pgpr_collect", &pgpr_collect);
    m.impl("oob_count", &oob_count);
}
