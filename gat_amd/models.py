"""`GAT` / `SpGAT` with the reference's `inference()` signature, plus the small part of
`BaseGAttN` the Cora drivers call (masked loss, masked accuracy, L2 + Adam training step).

    GAT.inference   — models/gat.py:7-31   (dense [B,N,N] additive mask)
    SpGAT.inference — models/sp_gat.py:7-34 (sparse adjacency, batch 1)

Composition (identical in both): layer 0 has n_heads[0] heads of width hid_units[0] and
residual=False; layers 1..L-1 have n_heads[i] heads and residual=residual; hidden layers
concatenate their heads; the output layer has n_heads[-1] heads of width nb_classes with the
identity activation, residual=False, and averages them.  in_drop=ffd_drop, coef_drop=attn_drop.
Each layer is ONE fused kernel sweep over all its heads instead of n_heads attn_head calls.

The reference calls these as plain functions on the class (no instance, no self) and lets
TensorFlow create variables on the way; here the variables come from the active VariableStore
(gat_amd.layers.variable_scope), created on the first call and reused afterwards.
"""
import torch

from . import layers, ops


class TFAdam(torch.optim.Optimizer):
    """tf.train.AdamOptimizer as the reference builds it (models/base_gattn.py:83: default beta1 = 0.9,
    beta2 = 0.999, epsilon = 1e-8).  TensorFlow's update is
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  var -= lr_t * m_t / (sqrt(v_t) + epsilon)
    i.e. epsilon joins the UNcorrected second-moment root, whereas torch.optim.Adam adds it after the
    bias correction (an epsilon sqrt(1 - beta2^t) times smaller in TensorFlow's terms: 32 times at the
    first step).  For gradients of the size an L2 penalty alone produces the two take visibly
    different steps, so the drop-in keeps TensorFlow's form."""

    def __init__(self, params, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-8, capturable=False, l2_coef=0.0):
        """capturable: keep the step count on the device and form lr_t there, so that step() issues
        the same launches with the same arguments every time and can be replayed from a HIP graph
        (gat_amd.graphs.capture); the host-side count of the default mode would be frozen at capture.
        l2_coef: add the gradient of the reference's penalty l2_coef * sum 0.5 ||v||^2
        (base_gattn.py:77-79), l2_coef * v, to every variable's gradient here, in one fused launch
        per few dozen variables, instead of through autograd (where it is a second contribution to
        every variable: one more accumulation launch each); the loss then carries the penalty's value
        only (BaseGAttN.training(..., penalty_in_optimizer=True))."""
        super().__init__(params, dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, capturable=capturable, l2_coef=l2_coef))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if not st:
                    st["t"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
            b1, b2 = group["beta1"], group["beta2"]
            if group["capturable"]:
                if "t_dev" not in group:
                    group["t_dev"] = torch.zeros((), dtype=torch.float64, device=ps[0].device)
                t = group["t_dev"]
                t.add_(1.0)
                lr_t = (group["lr"] * torch.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)).to(torch.float32)
            else:
                t = self.state[ps[0]]["t"] + 1
                for p in ps:
                    self.state[p]["t"] = t
                lr_t = group["lr"] * (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
            gs = [p.grad for p in ps]
            if group["l2_coef"]:
                gs = torch._foreach_add(gs, ps, alpha=group["l2_coef"])
            ms = [self.state[p]["m"] for p in ps]
            vs = [self.state[p]["v"] for p in ps]
            torch._foreach_mul_(ms, b1)
            torch._foreach_add_(ms, gs, alpha=1.0 - b1)
            torch._foreach_mul_(vs, b2)
            torch._foreach_addcmul_(vs, gs, gs, value=1.0 - b2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, group["eps"])
            if group["capturable"]:
                upd = torch._foreach_div(ms, den)
                torch._foreach_mul_(upd, lr_t)               # lr_t is a tensor on the device here
                torch._foreach_sub_(ps, upd)
            else:
                torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)
        return loss


class _L2Penalty(torch.autograd.Function):
    """coef * sum_v 0.5 ||v||^2 over a list of tensors with a handful of launches in each direction
    (horizontally fused norms forward, one fused scaling backward: d/dv = coef * v).  Written per
    tensor, the penalty of a 13-variable network is about a hundred launches of a few microseconds,
    a visible share of a Cora-sized training step even when the step is replayed from a HIP graph."""

    @staticmethod
    def forward(ctx, coef, *params):
        ctx.coef = coef
        ctx.save_for_backward(*params)
        sq = torch.stack(torch._foreach_norm([p.detach() for p in params])).double().square().sum()
        return (sq * (0.5 * coef)).to(params[0].dtype)

    @staticmethod
    def backward(ctx, grad):
        params = ctx.saved_tensors
        scale = grad * ctx.coef
        return (None, *torch._foreach_mul(list(params), scale))


class BaseGAttN:
    """Loss / metric / optimiser helpers of models/base_gattn.py."""

    @staticmethod
    def loss(logits, labels):
        """base_gattn.py:6-17: mean softmax cross-entropy against one-hot (or soft) labels."""
        return -(labels * torch.log_softmax(logits, dim=-1)).sum(-1).mean(0)

    @staticmethod
    def _fused_ce(logits, labels):
        return (logits.is_cuda and logits.dim() == 2 and logits.dtype == torch.float32 and labels.dtype == torch.float32
                and logits.shape[1] <= 256 and labels.shape == logits.shape)

    @staticmethod
    def masked_softmax_cross_entropy(logits, labels, mask):
        """base_gattn.py:140-146: per-node CE weighted by mask / mean(mask), mean over ALL nodes.  On the GPU one
        sweep over the logits gives the loss and its gradient (ops.masked_cross_entropy)."""
        if BaseGAttN._fused_ce(logits, labels):
            return ops.masked_cross_entropy(logits, labels, mask)[0]
        loss = -(labels * torch.log_softmax(logits, dim=-1)).sum(-1)
        mask = mask.to(loss.dtype)
        return (loss * (mask / mask.mean())).mean()

    @staticmethod
    def masked_accuracy(logits, labels, mask):
        """base_gattn.py:158-165."""
        if BaseGAttN._fused_ce(logits, labels):
            return ops.masked_cross_entropy(logits.detach(), labels, mask)[1]
        correct = (logits.argmax(1) == labels.argmax(1)).to(torch.float32)
        mask = mask.to(torch.float32)
        return (correct * (mask / mask.mean())).mean()

    @staticmethod
    def masked_loss_and_accuracy(logits, labels, mask):
        """The two quantities the drivers evaluate every epoch (execute_cora.py:74-75) from ONE sweep."""
        if BaseGAttN._fused_ce(logits, labels):
            return ops.masked_cross_entropy(logits, labels, mask)
        return BaseGAttN.masked_softmax_cross_entropy(logits, labels, mask), BaseGAttN.masked_accuracy(logits, labels, mask)

    @staticmethod
    def accuracy(preds, labels):
        """base_gattn.py:90-93."""
        return (preds.argmax(1) == labels.argmax(1)).to(torch.float32).mean()

    # ---- losses and metrics of the brain-network drivers (plain tensor arithmetic) -------------
    @staticmethod
    def _xent(logits, labels):
        return -(labels * torch.log_softmax(logits, dim=-1)).sum(-1).mean(0)

    @staticmethod
    def bi_loss(logits, labels, fmri_net, recon_net_fmri, DTI_net, recon_net_DTI, recon_lr_weight):
        """base_gattn.py:18-29: cross-entropy + weight * (MSE(fmri) + 0.1 * MSE(DTI))."""
        rec_f = ((fmri_net - recon_net_fmri) ** 2).mean()
        rec_d = ((DTI_net - recon_net_DTI) ** 2).mean()
        return BaseGAttN._xent(logits, labels) + recon_lr_weight * (rec_f + 0.1 * rec_d)

    @staticmethod
    def AE_classify_loss(logits, labels, fmri_net, recon_net_fmri, DTI_net, reconstruct_DTI, recon_lr_weight):
        """base_gattn.py:31-41: the DTI reconstruction error is weighted by |DTI_net|, the fMRI one is plain."""
        rec_d = (((DTI_net - reconstruct_DTI) ** 2) * DTI_net.abs()).mean()
        rec_f = ((fmri_net - recon_net_fmri) ** 2).mean()
        return BaseGAttN._xent(logits, labels) + recon_lr_weight * (rec_f + 0.1 * rec_d)

    @staticmethod
    def AE_regression_loss(logits, labels, fmri_net, recon_net_fmri, DTI_net, reconstruct_DTI, recon_lr_weight):
        """base_gattn.py:43-58: MSE on the target + weight * (|.|-weighted reconstruction errors)."""
        rec_d = (((DTI_net - reconstruct_DTI) ** 2) * DTI_net.abs()).mean()
        rec_f = (((fmri_net - recon_net_fmri) ** 2) * fmri_net.abs()).mean()
        return ((labels - logits) ** 2).mean() + recon_lr_weight * (rec_f + rec_d)

    @staticmethod
    def syn_loss(logits, labels, fmri_net, recon_net_fmri, recon_lr_weight):
        """base_gattn.py:61-72."""
        return ((labels - logits) ** 2).mean() + recon_lr_weight * ((fmri_net - recon_net_fmri) ** 2).mean()

    @staticmethod
    def syn_accuracy(preds, labels):
        """base_gattn.py:95-107: Pearson correlation of predictions and targets (epsilon 1e-7 as Keras')."""
        p, l = preds.squeeze(), labels.squeeze()
        pc, lc = p - p.mean(), l - l.mean()
        return (pc * lc).sum() / (torch.sqrt((pc * pc).sum() * (lc * lc).sum()) + 1e-7)

    @staticmethod
    def preshape(logits, labels, nb_classes):
        """base_gattn.py:109-114."""
        return logits.reshape(-1, nb_classes), labels.reshape(-1)

    @staticmethod
    def confmat(logits, labels):
        """base_gattn.py:116-118: rows = true class, columns = predicted class."""
        preds = logits.argmax(1)
        k = int(max(int(labels.max()), int(preds.max()))) + 1
        return torch.bincount(labels.long() * k + preds, minlength=k * k).reshape(k, k)

    @staticmethod
    def MSE(preds, labels):
        """base_gattn.py:121-124."""
        return ((labels.squeeze() - preds.squeeze()) ** 2).mean()

    @staticmethod
    def MAE(preds, labels):
        """base_gattn.py:126-129."""
        return (labels.squeeze() - preds.squeeze()).abs().mean()

    @staticmethod
    def micro_f1(logits, labels):
        """base_gattn.py:167-188 (two classes: the arg-max index doubles as the 0/1 prediction).
        Returns (f-measure, precision, recall)."""
        predicted, lab = logits.argmax(1), labels.argmax(1)
        tp = ((predicted * lab) != 0).sum().double()
        fp = ((predicted * (lab - 1)) != 0).sum().double()
        fn = (((predicted - 1) * lab) != 0).sum().double()
        precision, recall = tp / (tp + fp), tp / (tp + fn)
        return (2 * precision * recall / (precision + recall)).float(), precision, recall

    @staticmethod
    def l2_penalty(parameters, l2_coef):
        """base_gattn.py:77-79: l2_coef * sum of tf.nn.l2_loss(v) = 0.5 * ||v||^2 over every
        trainable variable (the name filter there compares against names no TF variable has, so
        biases are included too)."""
        return _L2Penalty.apply(float(l2_coef), *parameters)

    @staticmethod
    def training(parameters, lr, l2_coef, capturable=False, penalty_in_optimizer=False):
        """base_gattn.py:75-88 as (optimizer, penalty_fn).  Adam at a CONSTANT rate: the reference
        builds an exponential-decay schedule (:82) but never passes global_step to minimize (:86),
        so the step counter stays 0 and the rate never decays.  capturable: see TFAdam.
        penalty_in_optimizer: penalty_fn() returns the penalty's value without a gradient and the
        optimizer adds l2_coef * v to each gradient itself (the same update, a launch per variable
        fewer: worth it for networks of a hundred variables whose steps are a few milliseconds)."""
        parameters = list(parameters)
        if penalty_in_optimizer:
            opt = TFAdam(parameters, lr=lr, capturable=capturable, l2_coef=float(l2_coef))

            def value():
                with torch.no_grad():
                    return BaseGAttN.l2_penalty(parameters, l2_coef)
            return opt, value
        opt = TFAdam(parameters, lr=lr, capturable=capturable)
        return opt, (lambda: BaseGAttN.l2_penalty(parameters, l2_coef))


def _compose(head_fn, inputs, nb_classes, training, attn_drop, ffd_drop, graph, hid_units, n_heads,
             activation, residual):
    layers.default_store().begin()
    h = head_fn(inputs, hid_units[0], graph, activation, n_heads[0], "concat", ffd_drop, attn_drop, False, training)
    for i in range(1, len(hid_units)):
        h = head_fn(h, hid_units[i], graph, activation, n_heads[i], "concat", ffd_drop, attn_drop, residual, training)
    return head_fn(h, nb_classes, graph, "identity", n_heads[-1], "mean", ffd_drop, attn_drop, False, training)


def _is_training(training, attn_drop, ffd_drop):
    """Whether dropout acts.  As in the reference, `training` is accepted and NOT consulted
    (models/gat.py:8 never reads it): a rate other than 0 drops (utils/layers.py:8,19,21 test the
    rate itself), and evaluation feeds 0.0 for both rates (execute_cora.py:123)."""
    del training
    return attn_drop != 0.0 or ffd_drop != 0.0


class GAT(BaseGAttN):
    @staticmethod
    def inference(inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat, hid_units, n_heads, activation=layers.elu, residual=False):
        """models/gat.py:8-31.  inputs [B,N,F]; bias_mat [B,N,N] from adj_to_bias; returns [B,N,nb_classes]."""
        def head_fn(seq, out_sz, bm, act, nh, reduce, in_drop, coef_drop, res, train):
            return layers.attn_heads(seq, out_sz, bm, act, nh, reduce, in_drop, coef_drop, res, train)
        return _compose(head_fn, inputs, nb_classes, _is_training(training, attn_drop, ffd_drop), attn_drop, ffd_drop,
                        bias_mat, hid_units, n_heads, activation, residual)


class SpGAT(BaseGAttN):
    @staticmethod
    def inference(inputs, nb_classes, nb_nodes, training, attn_drop, ffd_drop,
                  bias_mat, hid_units, n_heads, activation=layers.elu, residual=False):
        """models/sp_gat.py:8-34.  inputs [1,N,F]; bias_mat is the sparse adjacency: the
        (indices, values, shape) triple of preprocess_adj_bias or a prebuilt CSRGraph."""
        graph = layers._graph_of(bias_mat, inputs.device)

        def head_fn(seq, out_sz, g, act, nh, reduce, in_drop, coef_drop, res, train):
            return layers.sp_attn_heads(seq, out_sz, g, act, nb_nodes, nh, reduce, in_drop, coef_drop, res, train)
        return _compose(head_fn, inputs, nb_classes, _is_training(training, attn_drop, ffd_drop), attn_drop, ffd_drop,
                        graph, hid_units, n_heads, activation, residual)


class GAT_AE_BNF_V2(torch.nn.Module):
    """models/gat.py:188-243 — the brain-network auto-encoder that BNF_AE.py:28 selects.  Two
    `reconstruction` branches (:194-216), each: batch-normalised connectivity (:201) ->
    attn_e2v_BNF_V2 with residual=True (:202) -> InnerProductDecoder (:204-206) and, after dropout
    with keep 0.7 in training (:209-210), conv1d(.,64,1) (:212); the two node embeddings are
    concatenated (:219), mapped to nb_CAM channels (:231), averaged over nodes (:233) and classified
    (:234); `cam` applies the classifier's kernel per node (:236-239).  TensorFlow builds variables
    per call; here they are created once for a fixed (f_in, nb_nodes).  The attention steps and the
    decoder run on the libgatmi kernels, the width-1 convolutions and the batch norm on torch."""

    def __init__(self, f_in, nb_nodes, nb_classes, nb_CAM, hid_units, n_heads, device=None):
        super().__init__()
        from .layers import AttnE2VBNFV2
        kw = {"device": device, "dtype": torch.float32}
        self.nb_nodes, self.nb_classes = nb_nodes, nb_classes
        # tf.contrib.layers.batch_norm defaults: statistics over all but the last axis, center only
        # (scale=False), epsilon 1e-3, decay 0.999, is_training=True
        self.bn = torch.nn.ModuleList([torch.nn.BatchNorm1d(nb_nodes, eps=1e-3, momentum=1e-3, affine=False, **kw) for _ in range(2)])
        self.bn_beta = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(nb_nodes, **kw)) for _ in range(2)])
        self.e2v = torch.nn.ModuleList([AttnE2VBNFV2(f_in, hid_units[0], n_heads[0], "relu", residual=True, device=device) for _ in range(2)])
        self.node_fc1 = torch.nn.ModuleList([torch.nn.Linear(hid_units[0], 64, **kw) for _ in range(2)])
        self.node_fc2 = torch.nn.Linear(128, nb_CAM, **kw)
        self.dense_out = torch.nn.Linear(nb_CAM, nb_classes, **kw)

    def _reconstruction(self, k, inputs, net_mat, training):
        B, N, _ = net_mat.shape
        nm = self.bn[k](net_mat.reshape(B * N, N)).reshape(B, N, N) + self.bn_beta[k]
        emb, _ = self.e2v[k](inputs, nm)
        graph = layers.InnerProductDecoder(emb).reshape(-1, N, N)
        if training:
            emb = torch.nn.functional.dropout(emb, 0.3, True)
        return layers._linear(self.node_fc1[k], emb, "relu"), graph

    def forward(self, inputs, net_mat, training=False):
        """inputs [B,N,F], net_mat [B,N,N] -> (digits [B,C], reconstruct_fmri [B,N,N],
        reconstruct_DTI [B,N,N], softmax(digits), cam [B,N,C]); the reference's `inference`
        returns the same five values."""
        fc_a, rec_a = self._reconstruction(0, inputs, net_mat, training)
        fc_b, rec_b = self._reconstruction(1, inputs, net_mat, training)
        node_fc2 = layers._linear(self.node_fc2, torch.cat([fc_a, fc_b], dim=-1), "relu")
        digits = self.dense_out(node_fc2.mean(dim=-2))
        cam = node_fc2 @ self.dense_out.weight.t()
        return digits, rec_a, rec_b, torch.softmax(digits, dim=-1), cam
