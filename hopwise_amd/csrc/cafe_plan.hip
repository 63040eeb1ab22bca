// The node tables of a MetaProgramExecutor batch built ON THE DEVICE (cafe.py:790-857: every user's program is the layout
// tree visited breadth-first with the children of the branching node in a random order, cafe.py:807-808).
//
// The layout tree is fixed per model, so a user's program is one of a few hundred visiting orders; the host keeps, per order,
// the program's columns (parent index, relation, depth, output capacity, leaf flag, leaf rows) as [n_orders, L] tables -- on
// the device too.  A call then needs from the host only each user's order index (its random draw): this kernel gathers the
// columns, numbers the (user, node) rows, lays out the output slots and writes the index columns the module stage reads --
// grouped by (depth, relation), users ascending inside a group, a user's nodes in program order: what one stable sort of the
// rows by that key gives.  Every order holds the same nodes, so a group's size is n_tasks x (nodes of that key per program)
// and its first row is known without a scan.  On the host the same took ~50 numpy passes over the n_tasks x L rows and a
// 2 MB copy per call: 0.9 ms for 1 024 users, 1.4 ms for 2 048 -- more than the device needs for the call's kernels once two
// calls overlap on two streams.
#include "hw_common.h"

namespace hw {

__global__ __launch_bounds__(256) void cafe_plan_kernel(
    const int64_t* __restrict__ oid, int64_t n_tasks, int L, int n_leaves, const int32_t* __restrict__ t_par,
    const int32_t* __restrict__ t_rel, const int32_t* __restrict__ t_depth, const int32_t* __restrict__ t_cap,
    const int32_t* __restrict__ t_is_leaf, const int32_t* __restrict__ t_leaf, int64_t n_orders,
    const int32_t* __restrict__ mod_rid, const int32_t* __restrict__ mod_eh, const int32_t* __restrict__ mod_vocab, int n_mods,
    const int32_t* __restrict__ key_off, const int32_t* __restrict__ key_cnt, int n_keys, int k, int64_t sum_cap,
    hw_cafe_node* __restrict__ nodes, int32_t* __restrict__ node_ptr, int64_t* __restrict__ slot_ptr,
    int64_t* __restrict__ eh_idx, int64_t* __restrict__ uv_idx, int64_t* __restrict__ row_idx,
    int32_t* __restrict__ leaf_nodes) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tasks * L) return;
    const int64_t b = t / L;
    const int pos = (int)(t - b * L);
    int64_t o = oid[b];
    o = o < 0 ? 0 : (o >= n_orders ? n_orders - 1 : o);  // the caller's draw indexes the table; clamp against garbage
    const int32_t* par_r = t_par + o * L;
    const int32_t* rel_r = t_rel + o * L;
    const int32_t* dep_r = t_depth + o * L;
    const int32_t* cap_r = t_cap + o * L;
    const int par = par_r[pos], rc = rel_r[pos], dep = dep_r[pos];
    int key = dep * n_mods + rc;
    key = key < 0 ? 0 : (key >= n_keys ? n_keys - 1 : key);
    int rank = 0;
    int64_t slot_off = 0;
    for (int p = 0; p < pos; ++p) {
        rank += (dep_r[p] * n_mods + rel_r[p]) == key;
        slot_off += cap_r[p];
    }
    const int64_t row = b * L + pos;
    const int64_t j = (int64_t)key_off[key] * n_tasks + b * key_cnt[key] + rank;
    eh_idx[j] = par < 0 ? b : n_tasks + b * L + par;
    uv_idx[j] = b;
    row_idx[j] = row;
    hw_cafe_node nd;
    nd.parent = par, nd.rel = mod_rid[rc], nd.eh_is_entity = mod_eh[rc], nd.k = k, nd.is_leaf = t_is_leaf[o * L + pos];
    nd.lp_row = (int32_t)row, nd.depth = dep, nd.vocab = mod_vocab[rc], nd.cap = cap_r[pos], nd.slot_off = (int32_t)slot_off;
    nodes[row] = nd;
    if (pos == 0) {
        node_ptr[b] = (int32_t)(b * L);
        slot_ptr[b] = b * sum_cap;
        if (b == n_tasks - 1) node_ptr[n_tasks] = (int32_t)(n_tasks * L), slot_ptr[n_tasks] = n_tasks * sum_cap;
    }
    for (int l = pos; l < n_leaves; l += L) leaf_nodes[b * n_leaves + l] = t_leaf[o * n_leaves + l];
}

}  // namespace hw

using namespace hw;

extern "C" int hw_cafe_plan(const int64_t* order_of_task, int64_t n_tasks, int32_t nodes_per_task, int32_t n_leaves,
                            const int32_t* t_parent, const int32_t* t_rel_code, const int32_t* t_depth, const int32_t* t_cap,
                            const int32_t* t_is_leaf, const int32_t* t_leaf_row, int64_t n_orders, const int32_t* mod_rid,
                            const int32_t* mod_eh_is_entity, const int32_t* mod_vocab, int32_t n_mods, const int32_t* key_first,
                            const int32_t* key_count, int32_t n_keys, int32_t k, int64_t slots_per_task, hw_cafe_node* nodes,
                            int32_t* node_ptr, int64_t* slot_ptr, int64_t* eh_idx, int64_t* uv_idx, int64_t* row_idx,
                            int32_t* leaf_nodes, void* stream) {
    if (!order_of_task || !t_parent || !t_rel_code || !t_depth || !t_cap || !t_is_leaf || !t_leaf_row || !mod_rid ||
        !mod_eh_is_entity || !mod_vocab || !key_first || !key_count || !nodes || !node_ptr || !slot_ptr || !eh_idx || !uv_idx ||
        !row_idx || !leaf_nodes)
        return HW_ERR_INVALID_ARG;
    if (n_tasks < 0 || nodes_per_task < 1 || n_leaves < 1 || n_orders < 1 || n_mods < 1 || n_keys < 1 || k < 1 ||
        slots_per_task < 0 || n_tasks * (int64_t)nodes_per_task > INT32_MAX)
        return HW_ERR_INVALID_ARG;
    if (n_tasks == 0) return HW_OK;
    HW_CLEAR_ERR();
    const int64_t total = n_tasks * nodes_per_task;
    hipLaunchKernelGGL(cafe_plan_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, order_of_task,
                       n_tasks, (int)nodes_per_task, (int)n_leaves, t_parent, t_rel_code, t_depth, t_cap, t_is_leaf, t_leaf_row,
                       n_orders, mod_rid, mod_eh_is_entity, mod_vocab, (int)n_mods, key_first, key_count, (int)n_keys, (int)k,
                       slots_per_task, nodes, node_ptr, slot_ptr, eh_idx, uv_idx, row_idx, leaf_nodes);
    return HW_LAUNCH_RC();
}
