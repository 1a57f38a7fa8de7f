#!/usr/bin/env python3
"""The brain-network auto-encoder the reference's BNF_AE.py selects (GAT_AE_BNF_V2, models/gat.py:188-243)
trained with its loss (AE_classify_loss, models/base_gattn.py:31-41) and Adam + L2
(base_gattn.py:75-88; BNF_AE.py:17-30 uses lr 0.001, l2 0.001, batch 50) on SYNTHETIC connectomes:
the HCP data the reference reads is restricted and absent.  Two classes of 90-node weighted
graphs differ in the strength of one block of connections; the node feature is the node's strength.

    python examples/train_bnf_ae_like.py --steps 60
    python examples/train_bnf_ae_like.py --steps 60 --replay     # the training step replayed from a HIP graph
    python examples/train_bnf_ae_like.py --nodes 246 --hid 64 --heads 6 --cam 256    # the sizes of BNF_AE.py:17-30,100

--replay: at the reference's batch of 50 a step is several hundred launches of a few microseconds each and the host
issuing them sets the pace; captured once (gat_amd.graphs.capture, Adam with its step count on the device) and
replayed on fresh data copied into the captured input tensors, the same step takes 3.1 ms instead of ≈ 9 ms (600 steps,
same loss curve; at the reference's sizes, second line above, 6.8 ms replayed and 10-13 ms eager).  The loop issues some twenty launches per step to draw its batch; gat_amd is imported before the first GPU
call so that the replay stays right through thousands of them (gat_amd/_hip_env.py).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_connectomes(batch, nodes, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    y = torch.randint(0, 2, (batch,), device=device, generator=g)
    w = torch.rand(batch, nodes, nodes, device=device, generator=g) * (torch.rand(batch, nodes, nodes, device=device, generator=g) < 0.3)
    k = nodes // 3
    w[:, :k, :k] = w[:, :k, :k] * (1.0 + 2.0 * y[:, None, None])      # class 1: a three times stronger block
    w = (w + w.transpose(1, 2)) / 2
    x = w.sum(-1, keepdim=True) / nodes
    return x, w, torch.nn.functional.one_hot(y, 2).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=50)
    ap.add_argument("--nodes", type=int, default=90)
    ap.add_argument("--hid", type=int, default=16, help="width of every attention round's output (BNF_AE.py:24 uses 64)")
    ap.add_argument("--heads", type=int, default=2, help="attention heads per round (BNF_AE.py:25 uses 6)")
    ap.add_argument("--cam", type=int, default=64, help="channels before the classifier (BNF_AE.py:100 uses 256)")
    ap.add_argument("--no-dropout", action="store_true", help="leave out the dropout of models/gat.py:209-210: eager and replayed runs then print the same losses")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--replay", action="store_true", help="capture the training step into a HIP graph and replay it")
    args = ap.parse_args()
    from gat_amd import graphs
    from gat_amd.models import BaseGAttN, GAT_AE_BNF_V2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = GAT_AE_BNF_V2(1, args.nodes, 2, nb_CAM=args.cam, hid_units=[args.hid], n_heads=[args.heads], device=dev)
    opt, l2 = BaseGAttN.training(model.parameters(), lr=0.001, l2_coef=0.001, capturable=args.replay, penalty_in_optimizer=True)
    xv, wv, yv = synthetic_connectomes(200, args.nodes, dev, 1)
    # the step works on these three tensors; every iteration copies its batch into them
    x, w, y = synthetic_connectomes(args.batch, args.nodes, dev, 100)
    loss_out = torch.zeros((), device=dev)

    def train_step():
        opt.zero_grad(set_to_none=True)       # under replay the gradients live in the graph's own pool, at fixed addresses
        digits, rec_a, rec_b, _, _ = model(x, w, training=not args.no_dropout)
        target = (w > 0).float()                                            # reconstruct which pairs are connected
        loss = BaseGAttN.AE_classify_loss(digits, y, target, rec_a, target, rec_b, 1.0) + l2()
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())

    # both modes start with two steps on the first batch: the warm-up calls of graphs.capture are real training
    # steps (the captured call itself only records its launches)
    done = 2
    if args.replay:
        train_step, _ = graphs.capture(train_step, warmup=2)
    else:
        for _ in range(done):
            train_step()
    first = last = None
    torch.cuda.synchronize()
    t0 = time.time()
    for step in range(done, args.steps):
        for dst, src in zip((x, w, y), synthetic_connectomes(args.batch, args.nodes, dev, 100 + step)):
            dst.copy_(src)
        train_step()
        if step % 10 == 0 or step == args.steps - 1:
            last = loss_out.item()
            first = last if first is None else first
            if not args.quiet and step % 10 == 0:
                print("step %3d  loss %.4f" % (step, last))
    torch.cuda.synchronize()
    per_step = (time.time() - t0) / max(1, args.steps - done) * 1e3
    with torch.no_grad():
        acc = float(BaseGAttN.accuracy(model(xv, wv, training=False)[0], yv))
    print("steps %d  %.2f ms per step%s  loss %.4f -> %.4f  held-out accuracy %.3f"
          % (args.steps, per_step, " (HIP-graph replay)" if args.replay else "", first, last, acc))
    return first, last, acc


if __name__ == "__main__":
    main()
