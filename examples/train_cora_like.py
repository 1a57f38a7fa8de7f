#!/usr/bin/env python3
"""Cora-style training loop on the MI355X kernels — the sparse driver of the reference
(execute_cora_sparse.py) with its hyper-parameters: hid_units=[8], n_heads=[8,1], lr 0.005,
l2 0.0005, dropout 0.6/0.6, ELU, early stopping with patience 100 on (val acc up or val loss down)
(execute_cora_sparse.py:16-26, 136-159).  The Planetoid files are not in the reference tree
(utils/process.py:50 reads ../data/), so the data is synthetic: a planted-partition citation-like
graph with Cora's shape (2708 nodes, 1433 sparse binary features, 7 classes).

    python examples/train_cora_like.py --epochs 200
    python examples/train_cora_like.py --epochs 200 --replay     # both steps replayed from HIP graphs

--replay captures the training step (forward with dropout, loss, backward, Adam) and the evaluation
step once and replays them: an eager step at this size is bound by the host issuing ~60 launches of
a few microseconds.  The in-kernel dropout masks change on every replay (gat_amd.graphs.capture keeps
a step word on the device for them) and Adam keeps its step count on the device (capturable).
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_cora(n=2708, f=1433, classes=7, seed=0):
    rng = np.random.RandomState(seed)
    labels = rng.randint(classes, size=n)
    # ~5300 undirected edges, 85 % inside a class (homophily like a citation graph)
    m = 5300
    a = rng.randint(n, size=m)
    same = rng.rand(m) < 0.85
    b = np.where(same, 0, rng.randint(n, size=m))
    for k in np.nonzero(same)[0]:
        cand = np.nonzero(labels == labels[a[k]])[0]
        b[k] = cand[rng.randint(len(cand))]
    adj = sp.coo_matrix((np.ones(m), (a, b)), shape=(n, n)).tocsr()
    adj = ((adj + adj.T) > 0).astype(np.float64)
    # class-dependent bag of words: each class prefers its own slice of the vocabulary
    feats = sp.lil_matrix((n, f))
    per = f // classes
    for i in range(n):
        own = rng.randint(labels[i] * per, (labels[i] + 1) * per, size=12)
        noise = rng.randint(f, size=8)
        feats[i, np.concatenate([own, noise])] = 1.0
    onehot = np.zeros((n, classes))
    onehot[np.arange(n), labels] = 1
    idx = rng.permutation(n)
    masks = [np.zeros(n, dtype=bool) for _ in range(3)]
    masks[0][idx[:140]] = True
    masks[1][idx[140:640]] = True
    masks[2][idx[640:1640]] = True
    return adj, feats.tocsr(), onehot, masks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--patience", type=int, default=100)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--sparse-features", action="store_true",
                    help="feed the sparse tuple of preprocess_features (stored entries only) instead of the dense matrix")
    ap.add_argument("--replay", action="store_true", help="capture the training and evaluation steps into HIP graphs and replay them")
    args = ap.parse_args()
    from gat_amd import CSRGraph, graphs, layers, preprocess
    from gat_amd.models import SpGAT

    dev = torch.device("cuda:0")
    adj, feats, onehot, (m_train, m_val, m_test) = synthetic_cora()
    features, features_tuple = preprocess.preprocess_features(feats)       # execute_cora_sparse.py:37
    biases = preprocess.preprocess_adj_bias(adj)                           # execute_cora_sparse.py:58
    graph = CSRGraph.from_reference_triple(biases, dev)                    # built once, reused every step
    n, f = features.shape
    if args.sparse_features:
        from gat_amd.features import SparseFeatures
        x = SparseFeatures.from_tuple(features_tuple, dev)                 # 1.3 % of the entries of the dense matrix
    else:
        x = torch.tensor(np.asarray(features)[None], dtype=torch.float32, device=dev)
    y = torch.tensor(onehot, dtype=torch.float32, device=dev)
    masks = {k: torch.tensor(v, device=dev) for k, v in (("train", m_train), ("val", m_val), ("test", m_test))}

    torch.manual_seed(0)
    store = layers.VariableStore()
    kw = dict(hid_units=[8], n_heads=[8, 1], activation=layers.elu, residual=False)

    def forward(train):
        with layers.variable_scope(store):
            drop = 0.6 if train else 0.0                                   # execute_cora_sparse.py:117-118, 133-134
            return SpGAT.inference(x, 7, n, train, drop, drop, bias_mat=graph, **kw)[0]

    forward(False)                                                         # creates the variables
    opt, l2 = SpGAT.training(store.parameters(), lr=0.005, l2_coef=0.0005, capturable=args.replay, penalty_in_optimizer=True)
    stats = torch.zeros(3, device=dev)                                     # train loss, val loss, val acc

    def train_step():
        opt.zero_grad(set_to_none=True)
        logits = forward(True)
        loss = SpGAT.masked_softmax_cross_entropy(logits, y, masks["train"]) + l2()
        loss.backward()
        opt.step()
        stats[0].copy_(loss.detach())

    def eval_step():
        with torch.no_grad():
            logits = forward(False)
            stats[1].copy_(SpGAT.masked_softmax_cross_entropy(logits, y, masks["val"]))
            stats[2].copy_(SpGAT.masked_accuracy(logits, y, masks["val"]))

    if args.replay:
        # the warm-up calls of the capture are real training steps: start the count from them
        state0 = {k: v.detach().clone() for k, v in store.state_dict().items()}
        train_step, _ = graphs.capture(train_step, warmup=2)
        eval_step, _ = graphs.capture(eval_step, warmup=1)
        store.load_state_dict(state0)
        for st in opt.state.values():
            st["m"].zero_()
            st["v"].zero_()
        for group in opt.param_groups:
            group["t_dev"].zero_()
    best_acc, best_loss, wait, best_state = 0.0, float("inf"), 0, None
    torch.cuda.synchronize()
    t0 = time.time()
    for epoch in range(args.epochs):
        train_step()
        eval_step()
        tl, vl, va = stats.tolist()
        if not args.quiet and epoch % 10 == 0:
            print("epoch %4d  train loss %.4f | val loss %.4f acc %.4f" % (epoch, tl, vl, va))
        if va >= best_acc or vl <= best_loss:                              # execute_cora_sparse.py:146-156
            if va >= best_acc and vl <= best_loss:
                best_state = {k: v.detach().clone() for k, v in store.state_dict().items()}
            best_acc, best_loss, wait = max(va, best_acc), min(vl, best_loss), 0
        else:
            wait += 1
            if wait == args.patience:
                break
    if best_state is not None:
        store.load_state_dict(best_state)
    with torch.no_grad():
        logits = forward(False)
        ta = SpGAT.masked_accuracy(logits, y, masks["test"]).item()
    print("epochs %d  %.2f s  best val acc %.4f  test acc %.4f" % (epoch + 1, time.time() - t0, best_acc, ta))
    return ta


if __name__ == "__main__":
    main()
