"""Build the id spaces and graph arrays the engine consumes from hopwise / RecBole atomic files
(`<name>.inter`, `<name>.kg`, `<name>.link`, optional `<name>.useremb / .entityemb / .relationemb`).

Restates, with numpy only, the part of the reference's KnowledgeBasedDataset that defines ids
(/root/reference/hopwise/data/dataset):
  _filter_link            kg_dataset.py:351-371   links whose item never interacts or whose entity never occurs in the
                                                  KG are dropped, then interactions with unlinked items are dropped
  _remap / _remap_ID_all  dataset.py:1036-1087    ids = order of first appearance (pandas.factorize) + 1, 0 = [PAD];
                                                  entity ids are factorised over head column then tail column
  _merge_item_and_entity  kg_dataset.py:556-588   unlinked items first, linked items last; a linked entity takes its
                                                  item's id; the remaining entities follow the items
  _add_auxiliary_relation kg_dataset.py:590-631   reverse relation r + n_rel with token r + "_r" (kg_reverse_r),
                                                  [UI-Relation] appended last
Pinned by the reference's own fixtures tests/data/{kg_remap_id,kg_reverse_r} (copied as data under
tests/golden/atomic) and the expectations of its tests/data/test_dataset.py:865-893.

The result quacks like the dataset object the model classes read: num(field), field2token_id, ui_relation, inter_feat,
head_entities / relations / tail_entities, history_item_matrix(), get_preload_weight().
"""

from __future__ import annotations

import os

import numpy as np


def read_atomic(path):
    """TSV with a `name:type` header -> dict name -> list of str (token / float kept as text)."""
    with open(path, encoding="utf-8") as f:
        header = f.readline().rstrip("\n").split("\t")
        names = [h.split(":")[0] for h in header]
        cols = {n: [] for n in names}
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            parts = line.split("\t")
            for n, p in zip(names, parts):
                cols[n].append(p)
    return cols


def factorize(tokens):
    """pandas.factorize: codes in order of first appearance, and the unique tokens in that order.  One dictionary pass (sorting
    a million Python strings through np.unique took half of the 6.8 s an ml-1m-sized dataset needed to load)."""
    table = {}
    codes = np.fromiter((table.setdefault(t, len(table)) for t in tokens), np.int64, len(tokens))
    uniq = np.empty(len(table), dtype=object)
    uniq[:] = list(table)
    return codes, uniq


class KGFiles:
    ui_relation = "[UI-Relation]"
    uid_field, iid_field = "user_id", "item_id"
    head_entity_field, tail_entity_field, relation_field, entity_field = "head_id", "tail_id", "relation_id", "entity_id"

    def __init__(self, data_path, name, kg_reverse_r=False, ui_relation="[UI-Relation]"):
        self.ui_relation = ui_relation
        base = os.path.join(data_path, name, name)
        inter = read_atomic(base + ".inter")
        kg = read_atomic(base + ".kg")
        link = read_atomic(base + ".link")
        users = np.asarray(inter[self.uid_field], dtype=object)
        items = np.asarray(inter[self.iid_field], dtype=object)
        heads = np.asarray(kg[self.head_entity_field], dtype=object)
        tails = np.asarray(kg[self.tail_entity_field], dtype=object)
        rels = np.asarray(kg[self.relation_field], dtype=object)

        # ---- _filter_link
        item2entity = dict(zip(link[self.iid_field], link[self.entity_field]))
        entity2item = {e: i for i, e in item2entity.items()}
        item_tokens, ent_tokens = set(items.tolist()), set(heads.tolist()) | set(tails.tolist())
        bad_items = {i for i, e in item2entity.items() if i not in item_tokens or e not in ent_tokens}
        bad_ents = {item2entity[i] for i in bad_items}
        for i in bad_items:
            del item2entity[i]
        for e in bad_ents:
            entity2item.pop(e, None)
        keep = np.array([i in item2entity for i in items], bool)
        users, items = users[keep], items[keep]

        # ---- _remap_ID_all (before the merge)
        u_code, u_tok = factorize(users)
        i_code, i_tok = factorize(items)
        e_code, e_tok = factorize(np.concatenate([heads, tails]))
        r_code, r_tok = factorize(rels)
        pad = np.array(["[PAD]"], dtype=object)
        user_token = np.concatenate([pad, u_tok])
        item_token = np.concatenate([pad, i_tok])
        entity_token = np.concatenate([pad, e_tok])
        rel_token = np.concatenate([pad, r_tok])
        u_id, i_id = u_code + 1, i_code + 1
        h_id, t_id = e_code[: len(heads)] + 1, e_code[len(heads):] + 1
        r_id = r_code + 1

        # ---- _merge_item_and_entity
        item_num, link_num, entity_num = len(item_token), len(item2entity), len(entity_token)
        item_priority = np.array([t in item2entity for t in item_token])
        item_order = np.argsort(item_priority, kind="stable")
        item_id_map = np.zeros_like(item_order)
        item_id_map[item_order] = np.arange(item_num)
        new_item_id2token = item_token[item_order]
        new_item_token2id = {t: i for i, t in enumerate(new_item_id2token)}
        i_id = item_id_map[i_id]
        entity_priority = np.array([t != "[PAD]" and t not in entity2item for t in entity_token])
        entity_order = np.argsort(entity_priority, kind="stable")
        entity_id_map = np.zeros_like(entity_order)
        for i in entity_order[1 : link_num + 1]:
            entity_id_map[i] = new_item_token2id[entity2item[entity_token[i]]]
        entity_id_map[entity_order[link_num + 1:]] = np.arange(item_num, item_num + entity_num - link_num - 1)
        new_entity_id2token = np.concatenate([new_item_id2token, entity_token[entity_order[link_num + 1:]]])
        for i in range(item_num - link_num, item_num):
            new_entity_id2token[i] = item2entity[new_entity_id2token[i]]
        h_id, t_id = entity_id_map[h_id], entity_id_map[t_id]

        # ---- _add_auxiliary_relation
        rel_token = list(rel_token)
        if kg_reverse_r:
            n_orig = len(rel_token) - 1
            for i in range(1, n_orig + 1):
                if rel_token[i] != self.ui_relation:
                    rel_token.append(rel_token[i] + "_r")
            h_id, t_id, r_id = (np.concatenate([h_id, t_id]), np.concatenate([t_id, h_id]),
                                np.concatenate([r_id, r_id + n_orig]))
        if self.ui_relation not in rel_token:
            rel_token.append(self.ui_relation)

        self.field2id_token = {
            self.uid_field: user_token, self.iid_field: new_item_id2token, self.entity_field: new_entity_id2token,
            self.head_entity_field: new_entity_id2token, self.tail_entity_field: new_entity_id2token,
            self.relation_field: np.array(rel_token, dtype=object),
        }
        self.field2token_id = {f: {t: i for i, t in enumerate(toks)} for f, toks in self.field2id_token.items()}
        # with kg_reverse_r the reference keeps duplicate-free tokens; relation_num counts the id space, which the
        # reverse block extends to 2 * n_orig + 1 even when a reverse token was skipped
        self.inter_feat = {self.uid_field: u_id.astype(np.int64), self.iid_field: i_id.astype(np.int64)}
        self.head_entities, self.tail_entities = h_id.astype(np.int64), t_id.astype(np.int64)
        self.relations = r_id.astype(np.int64)
        self.user_num, self.item_num = len(user_token), item_num
        self.entity_num, self.relation_num = len(new_entity_id2token), len(rel_token)
        self.item2entity, self.entity2item = item2entity, entity2item
        self._preload = {}
        for suffix, field, idf in (("useremb", "user_embedding", self.uid_field),
                                   ("entityemb", "entity_embedding", self.entity_field),
                                   ("relationemb", "relation_embedding", self.relation_field)):
            p = f"{base}.{suffix}"
            if os.path.exists(p):
                self._preload[field + "_id"] = self._load_embedding(p, field, self.field2token_id[idf])

    def _load_embedding(self, path, value_col, token2id):
        cols = read_atomic(path)
        id_col = next(c for c in cols if c.endswith("_id"))
        dim = len(cols[value_col][0].split(" "))
        tab = np.zeros((len(token2id), dim), np.float64)
        for tok, vec in zip(cols[id_col], cols[value_col]):
            if tok in token2id:
                tab[token2id[tok]] = np.array(vec.split(" "), np.float64)
        return tab

    # ---- the dataset protocol the models read -----------------------------------------------------------------
    def num(self, field):
        return {self.uid_field: self.user_num, self.iid_field: self.item_num, self.entity_field: self.entity_num,
                self.relation_field: self.relation_num}[field]

    def token2id(self, field, tokens):
        return np.array([self.field2token_id[field][t] for t in tokens])

    def get_preload_weight(self, name):
        return self._preload[name]

    def history_item_matrix(self):
        """[n_users, max_len] item ids of every user (0 padded), lengths -- dataset.py:1881-1985."""
        u, i = self.inter_feat[self.uid_field], self.inter_feat[self.iid_field]
        cnt = np.bincount(u, minlength=self.user_num)
        mat = np.zeros((self.user_num, max(int(cnt.max()), 1) if len(u) else 1), np.int64)
        pos = np.zeros(self.user_num, np.int64)
        for a, b in zip(u, i):
            mat[a, pos[a]] = b
            pos[a] += 1
        return mat, None, cnt
