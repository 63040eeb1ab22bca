"""CAFE inference (coarse-to-fine neural-symbolic reasoning) on the gfx950 kernels.

Reference: /root/reference/hopwise/model/knowledge_aware_recommender/cafe.py -- CAFE :26-480, SymbolicNetwork
:483-642, RelationModule :645-667, DeepRelationModule :670-699, KGMask :723-774, MetaProgramExecutor :777-867,
NeuralProgramLayout / TreeNode :870-964.

Same: ``CAFE(config, dataset)``; config keys (CAFE.yaml); one neural module per relation stored under
``model.<relation>.*`` (state_dict compatible); ``_infer_paths`` / ``run_program`` / ``full_sort_predict`` / ``explain``
semantics, including: the neighbour lookup uses the RELATION's head type, not the node's (cafe.py:603,823); visited ids
are per tree node; at executor leaves every pick joins the user's excluded set; ``execute`` keeps ``manual_topk = 5``
children per parent (adaptive_topk is False in run_program, cafe.py:790); result scores are the product of the three
log-probabilities (cafe.py:436).

Different by design: a user's whole program is interpreted by one workgroup (hw_cafe_execute) against
log-probability rows produced by the fp32 MFMA GEMM + a row log_softmax kernel, for all users of a batch at once;
the graph is a device CSR; nothing runs per user on the host except the final sort of a few hundred leaf paths.

Training (cafe.py:143-340): the replay memory and the bidirectional BFS path sampler stay host code over a host copy
of the CSR (as in the reference, where they walk a dict graph); ``SymbolicNetwork.forward`` runs on the device.
"""

from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .. import pgpr_ops as P
from ..lazy import LazyList
from ..stream_cache import FencedValue
from ..utils import InputType
from .abstract_recommender import ExplainableRecommender, KnowledgeRecommender


class _Packed:
    """Integer host arrays (int64 / int32, any shape) -> one int64 buffer -> ONE host-to-device copy -> typed views."""

    def __init__(self):
        self.parts, self.meta, self.off = [], {}, 0

    def add(self, name, arr):
        a = np.ascontiguousarray(arr)
        flat = a.reshape(-1)
        if a.dtype == np.int32:
            if flat.size & 1:
                flat = np.concatenate([flat, np.zeros(1, np.int32)])
            words = flat.view(np.int64)
        else:
            words = flat.astype(np.int64, copy=False)
        self.meta[name] = (self.off, words.size, a.dtype == np.int32, a.shape)
        self.parts.append(words)
        self.off += words.size

    # Pinned staging, four slots used in turn: a copy from pageable memory makes the launching thread wait until the stream
    # has drained (the previous call's kernels), which ties the host to the device once the host is the faster of the two; from
    # pinned memory the copy is one more stream operation.  A slot is rewritten only after the copy issued from it four calls
    # earlier has completed (its event).
    _SLOTS, _ring, _next = 4, {}, {}

    def to(self, dev):
        n = self.off
        if n == 0 or not torch.cuda.is_available() or torch.device(dev).type != "cuda":
            buf = torch.from_numpy(np.concatenate(self.parts) if self.parts else np.zeros(0, np.int64)).to(dev)
        else:
            key = torch.device(dev).index
            ring = _Packed._ring.setdefault(key, [None] * _Packed._SLOTS)
            i = _Packed._next.get(key, 0)
            _Packed._next[key] = (i + 1) % _Packed._SLOTS
            slot = ring[i]
            if slot is not None:
                slot[1].synchronize()
            if slot is None or slot[0].numel() < n:
                slot = ring[i] = [torch.empty(max(2 * n, 1 << 14), dtype=torch.int64, pin_memory=True), torch.cuda.Event()]
            np.concatenate(self.parts, out=slot[0].numpy()[:n])
            buf = torch.empty(n, dtype=torch.int64, device=dev)
            buf.copy_(slot[0][:n], non_blocking=True)
            slot[1].record()
        out = {}
        for name, (off, n, is32, shape) in self.meta.items():
            t = buf[off : off + n]
            if is32:
                t = t.view(torch.int32)[: int(np.prod(shape))]
            out[name] = t.view(shape) if len(shape) != 1 else t
        return out


class RelationModule(nn.Module):
    def __init__(self, embedding_size, info):
        super().__init__()
        self.name, self.eh_name, self.et_name = info["name"], info["entity_head"], info["entity_tail"]
        self.fc1 = nn.Linear(embedding_size * 2, 256)
        self.bn1 = ops.BatchNorm1d(256)
        self.fc2 = nn.Linear(256, embedding_size)
        self.dropout = ops.Dropout(0.5, 2020, stream="cafe." + info["name"])

    def forward(self, inputs):  # cafe.py:656-667
        eh_vec, user_vec = inputs
        x = torch.cat([eh_vec, user_vec], dim=-1)
        x = self.bn1(self.dropout(ops.linear_act(x, self.fc1.weight, self.fc1.bias, 2)))
        return ops.linear_act(x, self.fc2.weight, self.fc2.bias) + eh_vec


class DeepRelationModule(nn.Module):
    def __init__(self, embedding_size, info, use_dropout):
        super().__init__()
        self.name, self.eh_name, self.et_name = info["name"], info["entity_head"], info["entity_tail"]
        n = embedding_size * 2
        self.fc1 = nn.Linear(n, 256)
        self.bn1 = ops.BatchNorm1d(256)
        self.fc2 = nn.Linear(256, n)
        self.bn2 = ops.BatchNorm1d(n)
        self.fc3 = nn.Linear(n, embedding_size)
        self.dropout = ops.Dropout(0.5, 2020, stream="cafe." + info["name"]) if use_dropout else None

    def forward(self, inputs):  # cafe.py:686-699; the three Linear layers run on the MFMA GEMM
        eh_vec, user_vec = inputs
        feature = torch.cat([eh_vec, user_vec], dim=-1).contiguous()
        x = ops.linear_act(feature, self.fc1.weight, self.fc1.bias, 2)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.bn1(x)
        x = F.relu(ops.linear_act(x, self.fc2.weight, self.fc2.bias) + feature)
        if self.dropout is not None:
            x = self.dropout(x)
        x = self.bn2(x)
        return ops.linear_act(x, self.fc3.weight, self.fc3.bias)


class SymbolicNetwork(nn.Module):
    def __init__(self, relation_info, embedding_size, deep_module, use_dropout):
        super().__init__()
        for name, info in relation_info.items():
            mod = DeepRelationModule(embedding_size, info, use_dropout) if deep_module else RelationModule(embedding_size, info)
            setattr(self, name, mod)  # relation names contain dots; the reference registers them the same way

    def forward(self, metapath, pos_paths, neg_pids, embeddings):
        """(reg_loss, rank_loss) of a batch of positive paths along ``metapath`` (cafe.py:526-555): the chain of relation
        modules from the user vector, a cross entropy per level against that level's node over its entity vocabulary,
        and sigmoid(logp[neg item] - logp[pos item]) on the last level.  The vocabulary products run on the MFMA GEMM
        (frozen tables: only dX is formed), softmax / CE / ranking and their gradient in csrc/cafe.hip."""
        modules = [getattr(self, relation) for relation, _ in metapath[1:]]
        user_vec = embeddings["user"][pos_paths[:, 0]]
        x, reg_loss, rank_loss = user_vec, 0, None
        for i, module in enumerate(modules):
            x = module((x, user_vec)).view(pos_paths.shape[0], -1)
            logits = ops.matmul_nt(x, embeddings[module.et_name], padded=True)  # level_loss takes a row pitch
            last = i == len(modules) - 1
            ce, rk = ops.level_loss(logits, pos_paths[:, i + 1], pos_paths[:, -1] if last else None,
                                    neg_pids if last else None)
            reg_loss = reg_loss + ce
            if last:
                rank_loss = rk
        return reg_loss, rank_loss


class TreeNode:
    def __init__(self, level, entity, relation):
        self.level, self.entity, self.relation = level, entity, relation
        self.parent, self.children, self.sample_size = None, {}, 0

    def get_children(self):
        return list(self.children.values())

    def has_children(self):
        return len(self.children) > 0


def build_layout(metapaths):
    """NeuralProgramLayout.__init__ (cafe.py:873-888): the prefix tree of the metapaths."""
    root = TreeNode(0, "user", None)
    for mp in metapaths:
        node = root
        for i in range(1, len(mp)):
            if mp[i] not in node.children:
                node.children[mp[i]] = TreeNode(i, mp[i][1], mp[i][0])
                node.children[mp[i]].parent = node
            node = node.children[mp[i]]
    return root


class CAFE(ops.DropoutCounters, KnowledgeRecommender, ExplainableRecommender):
    input_type = InputType.USERWISE
    other_parameter_name = ["_dropout_calls"]  # the relation modules' mask streams resume with the checkpoint

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        dev = torch.device(self.device)
        self.raw_metapaths = config["path_constraint"]
        self.rank_weight = config["rank_weight"]
        self.deep_module = config["deep_module"]
        self.use_dropout = config["use_dropout"]
        self.topk_candidates = config["topk_candidates"]
        self.sample_size = config["sample_size"]
        self.topk_paths = config["topk_paths"]
        # the vocabulary log-softmax of run_program from the logits product's epilogue (ops.gemm_lse) instead of a row pass over
        # the [nodes, vocabulary] matrix; HW_CAFE_FUSED_LSE=0: product + hw_log_softmax_rows (equal to fp32 rounding)
        self.fused_lse = os.environ.get("HW_CAFE_FUSED_LSE", "1") != "0"
        # run_program's node tables and index columns built on the device from each user's visiting order (hw_cafe_plan) instead
        # of ~50 numpy passes and a 2 MB copy per call; HW_CAFE_DEVICE_PLAN=0: the host construction (equal element for element)
        self.device_plan = os.environ.get("HW_CAFE_DEVICE_PLAN", "1") != "0"
        self.ui_relation = dataset.ui_relation
        self.ui_relation_id = dataset.field2token_id["relation_id"][dataset.ui_relation]
        ue = np.asarray(dataset.get_preload_weight("user_embedding_id"))
        ee = np.asarray(dataset.get_preload_weight("entity_embedding_id"))
        re_ = np.asarray(dataset.get_preload_weight("relation_embedding_id"))
        self.user_embedding = torch.from_numpy(ue).to(dev, torch.float32).contiguous()
        self.entity_embedding = torch.from_numpy(ee).to(dev, torch.float32).contiguous()
        self.relation_embedding = torch.from_numpy(re_).to(dev, torch.float32).contiguous()
        self.embedding_size = self.user_embedding.shape[1]
        self.embeddings = {"user": self.user_embedding, "entity": self.entity_embedding}
        self.relation2rid = dataset.field2token_id["relation_id"]
        self.rid2relation = {v: k for k, v in self.relation2rid.items()}
        self.positives = np.asarray(dataset.history_item_matrix()[0])

        inter = dataset.inter_feat
        self.graph = P.DeviceCKG(self.n_users, self.n_items, self.n_entities, self.ui_relation_id,
                                 np.asarray(inter[dataset.uid_field]), np.asarray(inter[dataset.iid_field]),
                                 np.asarray(dataset.head_entities), np.asarray(dataset.relations),
                                 np.asarray(dataset.tail_entities), dev, ui_bidirectional=False)

        self.relation_info = {}
        for name in self.rid2relation.values():
            if name == "[PAD]":
                continue
            if name == f"{dataset.ui_relation}_r":
                raise ValueError("The ui_relation name should not end with '_r'.")
            head = "user" if name == dataset.ui_relation else "entity"
            self.relation_info.setdefault(name, {"name": name, "entity_head": head, "entity_tail": "entity"})
        self.metapaths = [[(n[0], n[1]) for n in mp] for mp in self.raw_metapaths]
        self.model = SymbolicNetwork(self.relation_info, self.embedding_size, self.deep_module, self.use_dropout)
        self.rng = np.random.default_rng()
        self.path_max_user_trials = int(config["max_user_trials"]) if "max_user_trials" in config else 5
        self.mpath_ids = list(range(len(self.metapaths)))
        self.memory_size = 10000  # cafe.py:87
        self.replay_memory = {m: [] for m in self.mpath_ids}
        self._ue64, self._ee64, self._re64 = (np.asarray(x, np.float64) for x in (ue, ee, re_))
        self._topk_user_items = None
        self._vocab = {"user": self.n_users, "entity": self.n_entities}
        self._bit_words = (max(self.n_users, self.n_entities) + 31) // 32

    # ------------------------------------------------------------------------------------------------
    def _module(self, relation):
        # nn.Module.__getattr__ walks three dicts per call and this is asked per program node and per returned path
        # (~10^5 times per 256-user batch): keep a plain name -> module table
        tab = self.__dict__.get("_module_table")
        if tab is None:
            tab = {name: getattr(self.model, name) for name in self.relation_info}
            self.__dict__["_module_table"] = tab
        return tab[relation]

    def _excluded_bits(self, users_np, exclude, users_dev=None):
        """Bitset [B, words] of the ids excluded at leaves: the user's train positives without the 0 padding for the
        executor (cafe.py:421-422), nothing for infer_with_path -- there the reference subtracts a torch tensor from a
        set of ints (cafe.py:606-607), which removes nothing because tensor elements hash by identity.

        The bitsets of ALL users are built once (array-wise) and kept on the device when they fit 256 MB; a batch is then
        one index_select instead of a Python loop over its users."""
        dev = self.user_embedding.device
        if not exclude:
            return torch.zeros((len(users_np), self._bit_words), dtype=torch.int32, device=dev)
        table = self.__dict__.get("_excluded_table")
        if table is None and self.positives.shape[0] * self._bit_words * 4 <= (256 << 20):
            pos = np.asarray(self.positives, np.int64)
            rows = np.repeat(np.arange(pos.shape[0], dtype=np.int64), pos.shape[1])
            cols = pos.reshape(-1)
            ok = (cols > 0) & (cols < self._bit_words * 32)
            rows, cols = rows[ok], cols[ok]
            bits = np.zeros(pos.shape[0] * self._bit_words, np.uint32)
            np.bitwise_or.at(bits, rows * self._bit_words + (cols >> 5), np.uint32(1) << (cols & 31).astype(np.uint32))
            table = self.__dict__["_excluded_table"] = torch.from_numpy(
                bits.view(np.int32).reshape(pos.shape[0], self._bit_words)).to(dev)
        if table is not None:
            if users_dev is None:
                users_dev = torch.from_numpy(np.asarray(users_np, np.int64)).to(dev)
            return table.index_select(0, users_dev).contiguous()
        bits = np.zeros((len(users_np), self._bit_words), np.uint32)
        for i, u in enumerate(users_np):
            pos = self.positives[int(u)]
            pos = pos[pos != 0]
            pos = pos[(pos >= 0) & (pos < self._bit_words * 32)]
            np.bitwise_or.at(bits[i], pos >> 5, (np.uint32(1) << (pos & 31).astype(np.uint32)))
        return torch.from_numpy(bits.view(np.int32)).to(dev)

    def _transposed(self, w):
        """[in, out] copy of an nn.Linear weight for hw_cafe_modules_forward_fused, cached per (storage, version): made once
        per weight version (evaluation reads the same weights call after call; an optimizer step bumps the version)."""
        cache = self.__dict__.setdefault("_wt_cache", {})
        fv = cache.get(id(w))
        if fv is None:
            fv = cache[id(w)] = FencedValue()  # calls on several streams read the copy (stream_cache.py)
        return fv.get((w.data_ptr(), w._version), lambda: w.detach().t().contiguous())

    def _module_structs(self, mods, user_vec, transposed=False):
        """ctypes table of the relation modules' parameters for hw_cafe_modules_forward, or None when the C++ loop does not
        apply (training mode, gradients wanted, dropout patched by a test, non-fp32 / host tensors, embedding size % 4).
        Pointers are read per call (an optimizer step may have replaced storage); nothing is copied -- except with
        ``transposed`` (the fused form): w1 / w2 / w3 then address cached [in, out] copies (:meth:`_transposed`)."""
        from .._lib import CafeModule

        if (torch.is_grad_enabled() or self.training or not user_vec.is_cuda or user_vec.dtype != torch.float32
                or user_vec.shape[1] % 4 or os.environ.get("HW_CAFE_MODULES", "1") == "0"):
            return None
        # the table of the previous call is reused while every parameter / buffer is the same object at the same address and
        # version (evaluation calls run_program again and again on unchanged weights; building the table was 0.2 ms of a
        # 1.3 ms call).  Parameters are read through the modules' own dictionaries: nn.Module.__getattr__ costs ~1 us a name.
        per_mod = []
        for m in mods:
            if m.training:
                return None
            mm = m._modules
            deep = isinstance(m, DeepRelationModule)
            fc1, bn1, fc2 = mm["fc1"], mm["bn1"], mm["fc2"]
            ps = [fc1._parameters["weight"], fc1._parameters["bias"], bn1._parameters["weight"], bn1._parameters["bias"],
                  bn1._buffers["running_mean"], bn1._buffers["running_var"], fc2._parameters["weight"], fc2._parameters["bias"]]
            if deep:
                bn2, fc3 = mm["bn2"], mm["fc3"]
                ps += [bn2._parameters["weight"], bn2._parameters["bias"], bn2._buffers["running_mean"],
                       bn2._buffers["running_var"], fc3._parameters["weight"], fc3._parameters["bias"]]
            else:
                ps += [None] * 6
            per_mod.append((deep, ps, m))
        sig = tuple((id(p), p.data_ptr(), p._version) if p is not None else None for _, ps, _ in per_mod for p in ps)
        cache = self.__dict__.setdefault("_struct_cache", {})
        fv = cache.get(bool(transposed))
        if fv is None:
            fv = cache[bool(transposed)] = FencedValue()

        def build():
            arr, keep = (CafeModule * len(mods))(), []
            for i, (deep, ps, m) in enumerate(per_mod):
                if any(p is not None and (not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous())
                       for p in ps) or ps[4] is None:
                    return None
                for name, p in zip(("w1", "b1", "bn1_weight", "bn1_bias", "bn1_mean", "bn1_var", "w2", "b2", "bn2_weight",
                                    "bn2_bias", "bn2_mean", "bn2_var", "w3", "b3"), ps):
                    if transposed and p is not None and name in ("w1", "w2", "w3"):
                        p = self._transposed(p)
                        keep.append(p)
                    setattr(arr[i], name, p.data_ptr() if p is not None else None)
                arr[i].bn1_eps, arr[i].bn2_eps = float(m.bn1.eps), float(m.bn2.eps) if deep else 0.0
                arr[i].deep = int(deep)
            return arr, keep  # the transposed copies ride along: a reader on another stream is fenced against their build

        hit = fv.get(sig, build)
        return None if hit is None else hit[0]

    @torch.no_grad()
    def _run_programs(self, users, programs, ks, grow_excluded, exclude, raw=False, flat=None):
        """programs[b] = list of (parent index, relation name) in execution order for user b; ks[b][i] = k of node i.
        Returns per user a list of (count, ids [count, depth], vals [count, depth]) per node -- or, with ``raw``, the
        interpreter's device buffers and the node tables (what :meth:`run_program` hands to the device collector).

        Host work is array-wise over all (user, node) rows: the programs of a batch are flattened once into numpy columns
        (parent row, relation, depth, capacity, slot offset) and every level's module inputs are one index_select."""
        dev = self.user_embedding.device
        users_np = users.cpu().numpy()
        B = len(users_np)
        dev_plan = flat.get("device_plan") if flat is not None else None
        if dev_plan is not None:
            # run_program on the usual layout: the node table, the pointers and the module stage's index columns are built on
            # the device from each user's visiting order (hw_cafe_plan); the host sends 2 B integers and keeps what it needs
            # as closed forms (every program has L nodes and the same number of output slots)
            if not raw:
                raise ValueError("the device-built plan serves the raw form only (run_program_device): no host columns exist")
            oid, TD, k_all = dev_plan
            names, mods = TD["names"], TD["mods"]
            Lp = TD["L"]
            total = B * Lp
            pk = _Packed()
            pk.add("oid", np.asarray(oid, np.int64))
            pk.add("users", np.asarray(users_np, np.int64))
            sent = pk.to(dev)
            on_dev = P.cafe_plan(sent["oid"], TD, B, k_all)
            on_dev["users"] = sent["users"]
            plan = [(dep, j, B * c) for dep, j, c in TD["plan1"]]
            et_rows = [(TD["et"], range(total))]
            nodes, cap, depth, slot_off = range(total), None, None, None  # (len() only; the raw form returns no host columns)
            node_ptr = np.arange(B + 1, dtype=np.int64) * Lp
            slot_ptr = np.arange(B + 1, dtype=np.int64) * TD["slots_per_task"]
            d = self.user_embedding.shape[1]
            vec_all = torch.empty((max(total, 1), d), dtype=torch.float32, device=dev)
        else:
            if flat is not None:  # run_program: the per-program columns come concatenated from the per-order cache
                lens, par_i, rel_code, k_of, names = flat["lens"], flat["par_i"], flat["rel_code"], flat["k_of"], flat["names"]
            else:
                lens = np.fromiter((len(p) for p in programs), np.int64, B)
            node_ptr = np.zeros(B + 1, np.int64)
            np.cumsum(lens, out=node_ptr[1:])
            total = int(node_ptr[-1])
            row_b = np.repeat(np.arange(B), lens)
            row_i = np.arange(total) - node_ptr[row_b]
            if flat is None:
                par_i = np.fromiter((q[0] for p in programs for q in p), np.int64, total)
                rel_names = [q[1] for p in programs for q in p]
                k_of = np.fromiter((int(x) for kk in ks for x in kk), np.int64, total)
                names = sorted(set(rel_names))
                code = {n: j for j, n in enumerate(names)}
                rel_code = np.fromiter((code[n] for n in rel_names), np.int64, total)
            par_row = np.where(par_i < 0, -1, node_ptr[row_b] + par_i)
            mods = [self._module(n) for n in names]
            if flat is not None and "depth" in flat:  # run_program: per-order columns, gathered (see _order_tables)
                depth, cap, is_leaf = flat["depth"], flat["cap"], flat["is_leaf"]
            else:
                # parents precede children inside a program: depth, capacity and leaf flags by program position
                depth = np.ones(total, np.int64)
                cap = k_of.copy()
                for pos in range(int(lens.max()) if B else 0):
                    rows = np.nonzero((row_i == pos) & (par_i >= 0))[0]
                    depth[rows] = depth[par_row[rows]] + 1
                    cap[rows] = cap[par_row[rows]] * k_of[rows]
                is_leaf = np.ones(total, np.int64)
                is_leaf[par_row[par_row >= 0]] = 0
            cum = np.cumsum(cap)
            slot_ptr = np.concatenate([[0], cum[node_ptr[1:] - 1]]) if total else np.zeros(B + 1, np.int64)
            slot_off = cum - cap - slot_ptr[row_b]
            # 1) node vectors, level by level, one module call per (level, relation) over all users
            d = self.user_embedding.shape[1]
            # every node row is written by its (level, relation) group before anything reads it (the groups partition the nodes)
            vec_all = torch.empty((max(total, 1), d), dtype=torch.float32, device=dev)
            # every index column of stages 1 and 2 goes to the device in ONE copy (they were three pageable host-to-device copies
            # per (level, relation) pair -- ~45 per call, each a stall of the launching thread in a leg that is host-bound)
            # the (level, relation) groups in (level, relation) order, rows ascending inside a group: ONE stable sort by the
            # group key instead of a mask + nonzero per pair (~50 passes over the node columns per call, most of the host time
            # of a 1 024-user run_program)
            plan, c_eh, c_uv, c_row = [], [], [], []
            if total:
                n_mods = len(mods)
                key = depth * n_mods + rel_code
                order = np.argsort(key, kind="stable")
                sizes = np.bincount(key, minlength=(int(depth.max()) + 1) * n_mods)
                plan = [(int(k // n_mods), int(k % n_mods), int(n)) for k, n in enumerate(sizes.tolist()) if n]
                pr = par_row[order]
                c_eh, c_uv, c_row = [np.where(pr < 0, row_b[order], B + pr)], [row_b[order]], [order]
            # rows per tail-entity type, types in name order (integer codes: a string column per node and np.unique over it were
            # a millisecond of a 1 024-user call)
            et_rows = []
            if total:
                et_name_of_mod = [m.et_name for m in mods]
                used_mods = np.nonzero(sizes.reshape(-1, n_mods).sum(0))[0].tolist()
                ets = sorted({et_name_of_mod[j] for j in used_mods})
                if len(ets) == 1:
                    et_rows = [(ets[0], np.arange(total))]
                else:
                    et_code = np.array([ets.index(n) if n in ets else -1 for n in et_name_of_mod], np.int64)[rel_code]
                    et_rows = [(et, np.nonzero(et_code == e)[0]) for e, et in enumerate(ets)]
            # the interpreter's node table (host-only inputs: built here so that it rides in the same copy)
            nodes = np.zeros(total, P.CAFE_NODE_DTYPE)
            nodes["parent"], nodes["k"], nodes["is_leaf"], nodes["depth"] = par_i, k_of, is_leaf, depth
            nodes["rel"] = np.array([self.relation2rid[n] for n in names], np.int64)[rel_code] if total else 0
            nodes["eh_is_entity"] = np.array([int(m.eh_name == "entity") for m in mods], np.int64)[rel_code] if total else 0
            nodes["vocab"] = np.array([self._vocab[m.et_name] for m in mods], np.int64)[rel_code] if total else 0
            nodes["lp_row"], nodes["cap"], nodes["slot_off"] = np.arange(total), cap, slot_off
            # ONE host-to-device copy per call: the index columns of stages 1 and 2 (three columns over all groups, then the
            # stage-2 row lists), the user ids, the node table and its pointers, the collector's leaf rows.  Each used to be its
            # own pageable copy (~0.2 ms of the launching thread apiece, eight of them) in a leg that is bound by the host.
            pk = _Packed()
            pk.add("idx", np.concatenate(c_eh + c_uv + c_row + [rs for _, rs in et_rows]) if plan or et_rows else np.zeros(0, np.int64))
            pk.add("users", np.asarray(users_np, np.int64))
            pk.add("nodes", nodes.view(np.int32).reshape(-1, 10))
            pk.add("node_ptr", np.asarray(node_ptr, np.int32))
            pk.add("slot_ptr", np.asarray(slot_ptr, np.int64))
            if flat is not None and "leaf_nodes" in flat:
                pk.add("leaf_nodes", np.ascontiguousarray(flat["leaf_nodes"], np.int32))
            on_dev = pk.to(dev)
        idx_all, users_dev = on_dev["idx"], on_dev["users"]
        user_vec = self.user_embedding[users_dev]
        offs = np.cumsum([0] + [n for _, _, n in plan])
        n_idx = int(offs[-1])
        eh_all, uv_all, row_all = idx_all[:n_idx], idx_all[n_idx : 2 * n_idx], idx_all[2 * n_idx : 3 * n_idx]
        pos = 3 * n_idx

        def take(n):
            nonlocal pos
            pos += n
            return idx_all[pos - n : pos]

        # HW_CAFE_MODULES: 2 (default) = one fused launch per level (cafe_level_kernel: fp32 FMA chains, equal to the others to
        # fp32 rounding); 1 = the GEMM / batch-norm launches of the loop below issued from C++ (bit-identical to it); 0 = the
        # loop below
        mode = os.environ.get("HW_CAFE_MODULES", "2")
        structs, done = None, False
        if plan and mode == "2" and 2 * user_vec.shape[1] <= 256:
            structs = self._module_structs(mods, user_vec, transposed=True)
            if structs is not None:
                done = P.cafe_modules_forward_fused(structs, [j for _, j, _ in plan], offs, [dep for dep, _, _ in plan], user_vec,
                                                    eh_all, uv_all, row_all, vec_all)
        if not done:
            structs = self._module_structs(mods, user_vec) if plan else None
        if done:
            pass
        elif structs is not None:
            # evaluation, fp32 on the device: the whole (level, relation) loop from one host call (csrc/cafe_modules.hip: the
            # same GEMM / batch-norm launches, issued from C++; bit-identical to the loop below)
            P.cafe_modules_forward(structs, [j for _, j, _ in plan], offs, user_vec, eh_all, uv_all, row_all, vec_all)
        else:
            src, src_dep = user_vec, 1
            for g, (dep, j, n) in enumerate(plan):
                if dep != src_dep:
                    src, src_dep = torch.cat([user_vec, vec_all]), dep
                o = int(offs[g])
                eh = src.index_select(0, eh_all[o : o + n])
                uv = user_vec.index_select(0, uv_all[o : o + n])
                vec_all.index_copy_(0, row_all[o : o + n], mods[j]((eh, uv)).view(n, -1))
        # 2) log-probabilities of every node over its tail vocabulary: one GEMM per entity type + row log_softmax
        # rows at a 16-byte pitch: a vocabulary of 13 805 entities makes every second row of a contiguous matrix start off a
        # 16-byte boundary, and the product that writes it falls back to 4-byte stores (526 against 334 us for the 13 312 x
        # 13 805 logits of a 1 024-user call, tools/bench_cafe_logits.py); the product writes into lp, the row pass runs in place
        width = max(self.n_users, self.n_entities)
        lp = ops.padded_rows(max(total, 1), width, dev)
        row_lse = None
        for et, rs in et_rows:
            rs_dev = take(len(rs))
            vocab = self._vocab[str(et)]
            if len(rs) == total:
                # one tail vocabulary for every node (the usual layout): the product leaves each row's (maximum, log sum exp)
                # beside the logits and the interpreter takes (x - maximum) - log sum exp at the entries it reads -- the row
                # pass rewrote 735 MB per 1 024 users (0.27 ms of a 1.4 ms call) for the few hundred entries per row read
                done = ops.gemm_lse(vec_all, self.embeddings[str(et)], out=lp[:, :vocab]) if self.fused_lse else None
                if done is not None:
                    row_lse = done[1]
                else:
                    sc = ops.gemm(vec_all, self.embeddings[str(et)], out=lp[:, :vocab])  # [n, d] @ E^T
                    P.log_softmax_rows(sc, out=sc)
            else:
                sc = ops.gemm(vec_all.index_select(0, rs_dev), self.embeddings[str(et)], out=ops.padded_rows(len(rs), vocab, dev))
                lp[rs_dev, :vocab] = P.log_softmax_rows(sc, out=sc)
        # 3) interpret
        excl = self._excluded_bits(users_np, exclude, users_dev)
        ids, vals, counts, nodes_dev, node_ptr_dev, slot_ptr_dev = P.cafe_execute(
            self.graph, nodes, node_ptr, users_dev, lp, excl, grow_excluded, slot_ptr, int(slot_ptr[-1]), keep_tables=True,
            on_device=(on_dev["nodes"], on_dev["node_ptr"], on_dev["slot_ptr"]), row_lse=row_lse)
        if raw:
            return dict(ids=ids, vals=vals, counts=counts, nodes=nodes_dev, node_ptr=node_ptr_dev, slot_ptr=slot_ptr_dev,
                        cap=cap, node_ptr_np=node_ptr, leaf_nodes=on_dev.get("leaf_nodes"))
        ids, vals, counts = ids.cpu().numpy(), vals.cpu().numpy(), counts.cpu().numpy()
        out = []
        for b in range(B):
            per = []
            for r in range(int(node_ptr[b]), int(node_ptr[b + 1])):
                s0 = int(slot_ptr[b] + slot_off[r])
                c, dep = int(counts[r]), int(depth[r])
                per.append((c, ids[s0 : s0 + c, :dep], vals[s0 : s0 + c, :dep]))
            out.append(per)
        return out

    # -- SymbolicNetwork.infer_with_path for every (user, metapath) ------------------------------------------
    def _infer_paths(self, users, kg_mask=None):
        users = torch.as_tensor(users)
        users_np = users.cpu().numpy()
        sample_sizes = [self.topk_paths, 5, 1]
        predictions = {int(u): {} for u in users_np}
        # one task per (user, metapath): a chain program
        task_users, programs, ks, owner = [], [], [], []
        for u in users_np:
            for mpid, mp in enumerate(self.metapaths):
                task_users.append(int(u))
                programs.append([(i - 1, mp[i + 1][0]) for i in range(len(mp) - 1)])
                ks.append([sample_sizes[i] for i in range(len(mp) - 1)])
                owner.append((int(u), mpid))
        res = self._run_programs(torch.tensor(task_users), programs, ks, grow_excluded=False, exclude=False)
        for (u, mpid), per in zip(owner, res):
            if any(c == 0 for c, _, _ in per):  # an empty level aborts the whole metapath (cafe.py:586, :638-639)
                predictions[u][mpid] = []
                continue
            c, ids, vals = per[-1]
            predictions[u][mpid] = [([u] + [int(x) for x in ids[j]], [np.float32(x) for x in vals[j]]) for j in range(c)]
        return predictions

    def _estimate_path_count(self, users):
        """Only feeds sample sizes that execute() ignores (adaptive_topk=False, cafe.py:790); kept for API parity as a
        uniform count."""
        return {int(u): np.zeros(len(self.metapaths)) for u in torch.as_tensor(users).cpu().numpy()}

    # -- MetaProgramExecutor.execute + collect_results + run_program ----------------------------------------------
    def _layout(self):
        """The layout tree as integer tables (built once: it only depends on the metapaths): nodes in plain BFS order,
        children ids per node in insertion order (what TreeNode.get_children returns), the root's children."""
        lay = self.__dict__.get("_layout_tables")
        if lay is None:
            root = build_layout(self.metapaths)
            nodes, index, queue = [], {}, list(root.get_children())
            while queue:
                n = queue.pop(0)
                index[id(n)] = len(nodes)
                nodes.append(n)
                queue.extend(n.get_children())
            lay = self.__dict__["_layout_tables"] = dict(
                root=root, nodes=nodes, kids=[[index[id(c)] for c in n.get_children()] for n in nodes],
                root_kids=[index[id(c)] for c in root.get_children()], orders={})
        return lay

    def _bfs_orders(self, B):
        """The visiting orders of B consecutive users.  When exactly one node of the layout tree has more than one child
        (metapaths that share nothing but a prefix up to one branching point: every user's walk draws ONE permutation, of the
        same size, at the same place) the B draws are one ``rng.permuted`` call over B rows -- the same values and the same
        generator state as B successive ``rng.permutation(n)`` calls (checked once per model on a copy of the generator;
        numpy shuffles the rows one after the other with the routine ``permutation`` uses) -- and a user's order is a
        dictionary lookup by its permutation's code: the per-user numpy call was 0.7 ms of a 1.8 ms run_program call."""
        lay = self._layout()
        fast = lay.get("fast")
        if fast is None:
            multi = [len(k) for k in lay["kids"] if len(k) > 1]
            fast = lay["fast"] = dict(on=False)
            if len(multi) == 1 and isinstance(self.rng, np.random.Generator):
                n = multi[0]
                try:
                    g1, g2 = (np.random.Generator(type(self.rng.bit_generator)()) for _ in range(2))
                    g1.bit_generator.state = g2.bit_generator.state = self.rng.bit_generator.state
                    seq = np.stack([g1.permutation(n) for _ in range(16)])
                    bat = g2.permuted(np.tile(np.arange(n), (16, 1)), axis=1)
                    same = np.array_equal(seq, bat) and g1.bit_generator.state == g2.bit_generator.state
                except Exception:  # an exotic bit generator: the per-user walk stays
                    same = False
                if same:
                    fast.update(on=True, n=n, base=np.tile(np.arange(n), (1, 1)), weights=n ** np.arange(n, dtype=np.int64),
                                by_code={}, bitgen=type(self.rng.bit_generator))
        # the check above vouches for ONE kind of generator: a caller that swaps self.rng (the bench's parity leg installs an
        # identity "rng", tests install seeded ones) gets the batched draw only for that kind, the per-user walk otherwise
        if (not fast["on"] or B == 0 or not isinstance(self.rng, np.random.Generator)
                or type(self.rng.bit_generator) is not fast["bitgen"]):
            return [self._bfs_order() for _ in range(B)]
        n = fast["n"]
        perms = self.rng.permuted(np.tile(fast["base"], (B, 1)), axis=1)
        codes = (perms @ fast["weights"]).tolist()
        by_code, out = fast["by_code"], []
        for b, c in enumerate(codes):
            ent = by_code.get(c)
            if ent is None:
                row = perms[b]
                ent = by_code[c] = self._bfs_order(perm=lambda k, row=row: row)
            out.append(ent)
        return out

    def _derive_order(self, e, manual_topk, leaves, code):
        """What run_program needs of a visiting order beyond the order itself: relation codes, the program row of every
        leaf chain, the record capacity (product of k along each leaf's chain)."""
        if "rel_code" not in e:
            e["rel_code"] = np.fromiter((code[n] for n in e["rel_names"]), np.int64, len(e["rel_names"]))
            index_of_chain = {_chain_of(e["order"], i): i for i in range(len(e["order"]))}
            e["leaf_row"] = np.asarray([index_of_chain.get(chain, -1) for chain in leaves] or [-1], np.int32)
        if e.get("topk") != manual_topk:  # capacity of a node = product of k along its chain
            n = len(e["order"])
            cap, depth, is_leaf = np.ones(n, np.int64), np.ones(n, np.int64), np.ones(n, np.int64)
            for i, (par, _, _) in enumerate(e["order"]):
                cap[i] = manual_topk * (cap[par] if par >= 0 else 1)
                if par >= 0:
                    depth[i], is_leaf[par] = depth[par] + 1, 0
            sel = e["leaf_row"][e["leaf_row"] >= 0]
            e["topk"], e["max_rec"] = manual_topk, int(cap[sel].sum())
            e["cap"], e["depth"], e["is_leaf"] = cap, depth, is_leaf

    def _order_tables(self, B, manual_topk, leaves, code):
        """Single branching node with at most 7 children (<= 5 040 visiting orders): every order enumerated once, its columns
        stacked into [n_orders, L] tables; a batch is then ONE permuted draw (as in :meth:`_bfs_orders`), one searchsorted
        from permutation codes to table rows and a gather per column -- no per-user Python at all.  Returns (order index per
        user, tables) or None when the layout / generator does not qualify (the caller walks the users then)."""
        lay = self._layout()
        if "fast" not in lay:
            self._bfs_orders(0)  # decides lay["fast"]
        fast = lay["fast"]
        if (not fast["on"] or B == 0 or fast["n"] > 7 or not isinstance(self.rng, np.random.Generator)
                or type(self.rng.bit_generator) is not fast["bitgen"]):
            return None
        n = fast["n"]
        tabs = fast.setdefault("tables", {})
        T = tabs.get(manual_topk)
        if T is None:
            import itertools

            rows = np.asarray(list(itertools.permutations(range(n))), np.int64)
            codes = rows @ fast["weights"]
            ents = []
            for row, c in zip(rows, codes.tolist()):
                ent = fast["by_code"].get(c)
                if ent is None:
                    ent = fast["by_code"][c] = self._bfs_order(perm=lambda k, row=row: row)
                self._derive_order(ent, manual_topk, leaves, code)
                ents.append(ent)
            srt = np.argsort(codes)
            T = tabs[manual_topk] = dict(
                codes=codes[srt], par=np.stack([ents[i]["par_i"] for i in srt]), rel=np.stack([ents[i]["rel_code"] for i in srt]),
                leaf=np.stack([ents[i]["leaf_row"] for i in srt]), max_rec=np.asarray([ents[i]["max_rec"] for i in srt], np.int64),
                depth=np.stack([ents[i]["depth"] for i in srt]), cap=np.stack([ents[i]["cap"] for i in srt]),
                is_leaf=np.stack([ents[i]["is_leaf"] for i in srt]))
        perms = self.rng.permuted(np.tile(fast["base"], (B, 1)), axis=1)
        return np.searchsorted(T["codes"], perms @ fast["weights"]), T

    def _order_tables_dev(self, T, names):
        """Device copies of the per-order tables (:meth:`_order_tables`) and what hw_cafe_plan needs beside them: the modules'
        relation id / head type / tail vocabulary by relation code, and the (depth, relation) groups of ONE program -- every
        order visits the same nodes, so the groups' sizes per program, the output slots per program and the single tail
        vocabulary hold for all of them (checked here once; None when they do not: the caller builds on the host)."""
        dev = self.user_embedding.device
        cache = T.setdefault("dev", {})
        key = (dev.type, dev.index, id(names))
        if key in cache:
            return cache[key]
        out = None
        n_mods = len(names)
        mods = [self._module(n) for n in names]
        keys = T["depth"] * n_mods + T["rel"]
        caps = T["cap"].sum(1)
        same = (np.sort(keys, axis=1) == np.sort(keys[:1], axis=1)).all() and (caps == caps[0]).all()
        n_keys = (int(T["depth"].max()) + 1) * n_mods
        count = np.bincount(keys[0], minlength=n_keys)
        ets = sorted({mods[j].et_name for j in np.nonzero(count.reshape(-1, n_mods).sum(0))[0].tolist()})
        if same and len(ets) == 1 and int(caps[0]) * 1 < 2 ** 31 and T["par"].shape[1] <= 128:
            i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(dev)
            out = dict(
                names=names, mods=mods, L=int(T["par"].shape[1]), n_leaves=int(T["leaf"].shape[1]), n_orders=int(T["par"].shape[0]),
                n_mods=n_mods, n_keys=n_keys, slots_per_task=int(caps[0]), et=ets[0],
                plan1=[(int(k // n_mods), int(k % n_mods), int(c)) for k, c in enumerate(count.tolist()) if c],
                par=i32(T["par"]), rel=i32(T["rel"]), depth=i32(T["depth"]), cap=i32(T["cap"]), is_leaf=i32(T["is_leaf"]),
                leaf=i32(T["leaf"]), mod_rid=i32([self.relation2rid[n] for n in names]),
                mod_eh=i32([int(m.eh_name == "entity") for m in mods]), mod_vocab=i32([self._vocab[m.et_name] for m in mods]),
                key_first=i32(np.cumsum(count) - count), key_count=i32(count))
        cache[key] = out
        return out

    def _bfs_order(self, perm=None):
        """One visiting order of the executor (children of a node in self.rng.permutation order, cafe.py:807-808; the
        draws happen per call, exactly as before) -> its cache entry: the order as (parent index, relation, node) triples
        plus everything run_program derives from it (flattened program columns, leaf rows, record capacity).  ``perm``
        stands in for ``self.rng.permutation`` when the draw was made elsewhere (:meth:`_bfs_orders`)."""
        lay = self._layout()
        kids_of, nodes = lay["kids"], lay["nodes"]
        # the queue IS the visiting order (nodes are appended in the order they will be popped): two flat lists and a cursor --
        # this runs once per user of a batch and was half of run_program's host time as a pop(0) / tuple loop
        ids = list(lay["root_kids"])
        pars = [-1] * len(ids)
        if perm is None:
            perm = self.rng.permutation
        head = 0
        while head < len(ids):
            kids = kids_of[ids[head]]
            n = len(kids)
            if n > 1:
                ids.extend([kids[j] for j in perm(n).tolist()])
                pars.extend([head] * n)
            elif n:
                ids.append(kids[0])
                pars.append(head)
            head += 1
        key = tuple(ids)
        ent = lay["orders"].get(key)
        if ent is None:
            order = [(par, nodes[nid].relation, nodes[nid]) for nid, par in zip(ids, pars)]
            ent = lay["orders"][key] = dict(order=order, par_i=np.asarray(pars, np.int64),
                                            rel_names=[nodes[nid].relation for nid in ids])
        return ent

    def _bfs_program(self):
        """Nodes of the layout tree in the executor's BFS order as (parent index, relation, node) triples."""
        return self._bfs_order()["order"]

    def _leaf_chains(self):
        """Relation chains of the leaves in collect_results order: plain BFS over the layout tree (cafe.py:859-867)."""
        root = build_layout(self.metapaths)
        leaves, queue = [], root.get_children()
        while queue:
            node = queue.pop(0)
            queue.extend(node.get_children())
            if not node.has_children():
                chain, n = [], node
                while n.parent is not None:
                    chain.append(n.relation)
                    n = n.parent
                leaves.append(tuple(reversed(chain)))
        return leaves

    def run_program_device(self, users, manual_topk=5):
        """run_program with everything left on the device: (scores [B, n_items] float32, record tensors, leaf chains).
        Nothing here waits for the device: a user whose paths reach more distinct items than hw_cafe_collect ranks (8 192) gets NaN scores and
        rec["overflow"] holds the count (P.cafe_overflow_check(rec) raises on it; run_program does that after its copy).
        Per-user host work is one rng.permutation draw and one dictionary lookup: program columns, leaf rows and record
        capacities are cached per visiting order of the layout tree."""
        users = torch.as_tensor(users)
        users_np = users.cpu().numpy()
        B = len(users_np)
        leaves = self._leaf_chains_cached()
        names = self.__dict__.get("_all_rel_names")
        if names is None:
            names = self.__dict__["_all_rel_names"] = sorted(self.relation_info)
            self.__dict__["_rel_code_of"] = {n: j for j, n in enumerate(names)}
        code = self.__dict__["_rel_code_of"]
        got = self._order_tables(B, manual_topk, leaves, code)
        TD = None
        if got is not None and self.device_plan:
            TD = self._order_tables_dev(got[1], names)
        if TD is not None:  # the usual layout, tables on the device: the host keeps the order indices only
            oid, T = got
            flat = dict(device_plan=(oid, TD, manual_topk), names=names)
            leaf_nodes = T["leaf"][oid]
            max_rec_all = int(T["max_rec"][oid].max())
        elif got is not None:  # the usual layout: columns gathered from the per-order tables
            oid, T = got
            flat = dict(lens=np.full(B, T["par"].shape[1], np.int64), par_i=T["par"][oid].ravel(),
                        rel_code=T["rel"][oid].ravel(), names=names, depth=T["depth"][oid].ravel(),
                        cap=T["cap"][oid].ravel(), is_leaf=T["is_leaf"][oid].ravel())
            leaf_nodes = T["leaf"][oid]
            max_rec_all = int(T["max_rec"][oid].max())
        else:
            ents = self._bfs_orders(B)
            for e in {id(e): e for e in ents}.values():
                self._derive_order(e, manual_topk, leaves, code)
            flat = dict(lens=np.fromiter((len(e["par_i"]) for e in ents), np.int64, B),
                        par_i=np.concatenate([e["par_i"] for e in ents]) if B else np.zeros(0, np.int64),
                        rel_code=np.concatenate([e["rel_code"] for e in ents]) if B else np.zeros(0, np.int64), names=names)
            leaf_nodes = np.stack([e["leaf_row"] for e in ents]) if B else np.full((0, max(len(leaves), 1)), -1, np.int32)
            max_rec_all = max([1] + [e["max_rec"] for e in ents])
        if TD is None:
            flat["k_of"] = np.full(len(flat["par_i"]), manual_topk, np.int64)
            flat["leaf_nodes"] = leaf_nodes  # rides in _run_programs' one host-to-device copy
        raw = self._run_programs(users, None, None, grow_excluded=True, exclude=True, raw=True, flat=flat)
        max_rec = min(max(1, max_rec_all), self.n_items)
        scores, rec = P.cafe_collect(raw, leaf_nodes, self.n_items, max_rec, check_overflow=False,
                                     leaf_nodes_dev=raw.get("leaf_nodes"))
        return scores, rec, leaves, users_np

    def _leaf_chains_cached(self):
        lc = self.__dict__.get("_leaf_chain_cache")
        if lc is None:
            lc = self.__dict__["_leaf_chain_cache"] = self._leaf_chains()
        return lc

    def run_program(self, users, path_counts=None, predicted_paths=None, manual_topk=5):
        """(scores [B, n_items], [[user, item, score, path], ...]) -- cafe.py:406-447.  Paths carry relation NAMES.

        collect_results and the ranking loop (cafe.py:417-441: every leaf path scored by the product of its float32
        log-probs, inst[item] = the LAST path to an item, stable sort by score, first entry per item emitted with inst[item]'s
        path) run on the device (hw_cafe_collect, one workgroup per user).  The second return value is a list-like view of
        the device records: the Python lists the reference's contract asks for are only built when somebody reads them
        (the ranking metrics of an evaluation never do)."""
        scores, rec, leaves, users_np = self.run_program_device(users, manual_topk)
        scores = scores.cpu()
        P.cafe_overflow_check(rec)  # after the copy: the device work is over, the read costs nothing more
        return scores, _LazyRecords(self, rec, leaves, users_np)

    def _records_to_lists(self, rec, leaves, users_np):
        cnt = rec["count"].cpu().numpy()
        item, score, leaf, pid = (rec[k].cpu().numpy() for k in ("item", "score", "leaf", "ids"))
        # (relation name, tail type) chains of the leaves, and the records as plain Python values in one pass each
        chains = [tuple((rel, self._module(rel).et_name) for rel in chain) for chain in leaves]
        collected = []
        for b in range(len(users_np)):
            c = int(cnt[b])
            if not c:
                continue
            u = int(users_np[b])
            start = ("self_loop", "user", u)
            sc_b = score[b, :c]
            rows = zip(item[b, :c].tolist(), sc_b, leaf[b, :c].tolist(), pid[b, :c].tolist())
            # explicit tuples per chain length: a generator per record costs more than the record itself
            collected.extend(
                [u, it, sc, ([start, (ch[0][0], ch[0][1], p[0]), (ch[1][0], ch[1][1], p[1]), (ch[2][0], ch[2][1], p[2])]
                             if len(ch) == 3 else [start, *((rel, et, e) for (rel, et), e in zip(ch, p))])]
                for it, sc, l, p in rows for ch in (chains[l],))
        return collected

    def full_sort_predict(self, interaction):
        scores, paths = self.run_program(interaction[self.USER_ID])
        # convert_path_relations (cafe.py:341-348): relation names -> ids -- applied when somebody reads the records
        rid = self.relation2rid
        conv = lambda rec: [rec[0], rec[1], rec[2], [rec[3][0]] + [(rid[r], t, e) for r, t, e in rec[3][1:]]]
        paths = paths.mapped(conv) if isinstance(paths, LazyList) else [conv(r) for r in paths]
        return scores, paths

    def explain(self, interaction):
        scores, explanations = self.run_program(interaction[self.USER_ID])
        if isinstance(explanations, LazyList):  # decoded when read: ExplainableCollector only keeps them for rec.paths metrics
            return scores, explanations.mapped(lambda e: e[:-1] + [self.decode_path(e[-1])])
        for e in explanations:
            e[-1] = self.decode_path(e[-1])
        return scores, explanations

    def decode_path(self, path):
        return path

    def predict(self, interaction):
        return

    # -- training (cafe.py:143-340) ----------------------------------------------------------------------------------
    def _compute_top_items(self):
        """cafe.py:134-141: the topk_candidates best items of every user under (U + R_ui) . E_item (float64, host, once)."""
        if self._topk_user_items is None:
            sc = np.dot(self._ue64 + self._re64[self.ui_relation_id], self._ee64[: self.n_items].T)
            order = np.argsort(sc, axis=1)[:, -100:][:, ::-1]
            self._topk_user_items = order[:, : self.topk_candidates]
        return self._topk_user_items

    def _rev_rel(self, rel):
        if rel == self.ui_relation:
            return self.ui_relation
        return rel[:-2] if rel.endswith("_r") else rel + "_r"

    def _nbrs(self, ntype, nid, relation_name):
        rid = self.relation2rid.get(relation_name)
        if rid is None:
            return np.zeros(0, np.int32)
        return P.host_neighbours(self.graph, ntype, nid, rid)

    def fast_sample_path_with_target(self, mpath_id, user_id, target_id, num_paths, sample_size=100):
        """Bidirectional BFS between the user and the target item along a metapath (cafe.py:207-289)."""
        metapath = self.metapaths[mpath_id]
        path_len = len(metapath) - 1
        mid = path_len // 2

        def cap(ids):
            return self.rng.choice(ids, size=sample_size, replace=False) if len(ids) > sample_size else ids

        forward_paths = [[int(user_id)]]
        for i in range(1, mid + 1):
            last_entity, next_relation = metapath[i - 1][1], metapath[i][0]
            forward_paths = [fp + [int(n)] for fp in forward_paths for n in cap(self._nbrs(last_entity, fp[-1], next_relation))]
        backward_paths = [[int(target_id)]]
        for i in reversed(range(mid + 2, path_len + 1)):
            next_relation, next_entity = metapath[i]
            backward_paths = [[int(c)] + bp for bp in backward_paths
                              for c in cap(self._nbrs(next_entity, bp[0], self._rev_rel(next_relation)))]
        backward_map = {}
        next_relation, next_entity = metapath[mid + 1]
        for bp in backward_paths:
            for c in cap(self._nbrs(next_entity, bp[0], self._rev_rel(next_relation))):
                backward_map.setdefault(int(c), []).append(bp)
        final_paths = []
        for fp_idx in self.rng.permutation(len(forward_paths)):
            fp = forward_paths[fp_idx]
            if fp[-1] not in backward_map:
                continue
            self.rng.shuffle(backward_map[fp[-1]])
            for bp in backward_map[fp[-1]]:
                final_paths.append(fp + bp)
                if len(final_paths) >= num_paths:
                    return final_paths
        return final_paths

    def _get_batch_by_user(self, users):
        """One (metapath, positive path, negative item) per user (cafe.py:143-198).  The metapath id returned is the LAST
        one drawn, as in the reference, which scores the whole batch with it."""
        top_items = self._compute_top_items()
        pos_paths, neg_pids = [], []
        trials = {int(u): 0 for u in users}
        top_pids = np.arange(self.topk_candidates)
        mpid, it = 0, 0
        while len(pos_paths) < len(users) and it < len(users):
            user = int(users[it])
            if trials[user] >= self.path_max_user_trials:
                it += 1
                continue
            mpid = int(self.rng.choice(self.mpath_ids))
            pidx = int(self.rng.choice(top_pids))
            item = int(top_items[user][pidx])
            memory = self.replay_memory[mpid]
            if self.rng.random() < 0.5 * len(memory) / self.memory_size:
                pos_paths.append(memory[int(self.rng.integers(len(memory)))])
            else:
                paths = self.fast_sample_path_with_target(mpid, user, item, 1)
                if not paths:
                    trials[user] += 1
                    continue
                pos_paths.append(paths[0])
                memory.extend(paths)
                del memory[: max(0, len(memory) - self.memory_size)]
            if pidx < self.topk_candidates - 1:
                neg_pids.append(int(top_items[user][int(self.rng.choice(np.arange(pidx + 1, self.topk_candidates)))]))
            else:
                neg_pids.append(int(self.rng.choice(self.n_items)))
            it += 1
        return mpid, np.array(pos_paths, np.int64), np.array(neg_pids, np.int64)

    def optimizer_step_groups(self):
        """Parameter lists that are always stepped together: one per relation module (a training step runs the modules of ONE
        metapath and the optimizer steps exactly their parameters).  optim.FusedAdam.pin_step_groups gives each a device step
        counter of its own, so that the captures of driver.GraphedStagedTrainStep never see the optimizer regroup."""
        return [list(mod.parameters()) for mod in self.model.children()]

    def graph_host_stage(self, interaction):
        """The host half of :meth:`calculate_loss` (driver.GraphedStagedTrainStep's protocol): the reference's sampler
        (cafe.py:143-198) -> (key, host tensors).  ``key`` = (metapath id, rows) names the device program that follows -- the
        chain of relation modules of that metapath on that many paths; the tensors are its inputs, still on the host."""
        # numpy, not torch, for the filter: a torch CPU op of this size opens an OpenMP region, which on a box whose CPU share is
        # smaller than its core count costs milliseconds next to the autograd thread (17 ms per step measured at 4 096 users)
        users = interaction[self.USER_ID].detach().cpu().numpy()
        users = users[users != 0]
        mpid, pos_paths, neg_pids = self._get_batch_by_user(users)
        return (int(mpid), int(len(pos_paths))), {"pos_paths": torch.from_numpy(pos_paths), "neg_pids": torch.from_numpy(neg_pids)}

    def graph_device_stage(self, key, tensors):
        """The device half: the symbolic network of metapath ``key[0]`` on the sampled batch -> (reg_loss, rank_weight *
        rank_loss), cafe.py:329-340."""
        reg_loss, rank_loss = self.model(self.metapaths[key[0]], tensors["pos_paths"], tensors["neg_pids"], self.embeddings)
        return reg_loss, rank_loss * self.rank_weight

    def calculate_loss(self, interaction):
        """(reg_loss, rank_weight * rank_loss) -- cafe.py:329-340; the trainer sums the tuple."""
        key, host = self.graph_host_stage(interaction)
        dev = self.user_embedding.device
        # through pinned staging blocks (torch's caching host allocator): a pageable source beyond ~64 KB takes the runtime's
        # pin-on-the-fly path, 10 ms per copy at 4 096 paths
        return self.graph_device_stage(key, {k: v.pin_memory().to(dev, non_blocking=True) for k, v in host.items()})


class _LazyRecords(LazyList):
    """The [[user, item, score, path], ...] list of run_program, materialised from the device records on first use (a real
    list afterwards: hopwise_amd/lazy.py)."""

    def __init__(self, model, rec, leaves, users_np):
        super().__init__(lambda: model._records_to_lists(rec, leaves, users_np))


def _chain_of(order, i):
    """Relation chain (root -> node) of node i of a BFS-ordered program."""
    chain = []
    while i >= 0:
        par, rel, _ = order[i]
        chain.append(rel)
        i = par
    return tuple(reversed(chain))
