"""Action-conditional contrastive predictive coding (`cpca`), the reference's one registered auxiliary loss
(habitat_baselines/rl/ppo/cpc_aux_loss.py:64-355; config `habitat_baselines.rl.auxiliary_losses.cpca`,
default_structured_configs.py:539-544).

Where it runs: `NetPolicy.evaluate_actions` on the autograd bridge hands every registered module
`aux_loss_state = {"rnn_output": [P, H], "perception_embed": [P, H]}` (outputs of the ENGINE's forward) and the minibatch; autograd's
gradients with respect to the two tensors re-enter the engine's backward through `hab_policy_set_extra_grads` (policy.py, _EvaluateFn).
The module itself is small host-launched torch work on [k, M, H] tensors (M <= time_subsample x number of sequences in the minibatch),
not part of the engine's parameter arena; its parameters are stepped by the updater's second Adam (ppo.py).

What is computed, for a minibatch of S packed sequences (rnn_build_seq_info of rnn_state_encoder.py:171-184):
  1. per sequence, up to `time_subsample` start steps (all of 1..len-1 when the sequence is short, else the head of a random permutation);
  2. from each start, the next k actions drive an LSTM whose initial (h, c) is the policy's recurrent output at the start step;
  3. of the k x M predictions, `future_subsample` per start are kept (see `_kept_predictions` for the exact rows);
  4. a two-layer head scores (prediction, perception embedding) pairs: the embedding one step ahead of the prediction is the positive,
     `num_negatives` embeddings drawn from the rest of the minibatch are negatives; binary cross-entropy, scaled by `loss_scale`.
The random draws (permutations, the kept futures, the negatives) are taken from torch's global generator of the tensors' device in the
reference's order, so a seeded run on one device reproduces the reference's loss value (tests/test_host_logic.py pins this on CPU
against the live reference).  Parameter names follow the reference (`_action_embed`, `_future_predictor`, `_predictor_first_layers`,
`_predictor`) for checkpoint interchange.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from habitat_amd.common.baseline_registry import baseline_registry
from habitat_amd.rl.models.action_embedding import ActionEmbedding


def masked_mean(t: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """cpc_aux_loss.py:25-31.  NOTE what this is: mean over ALL entries of t with the invalid ones zeroed, times the valid fraction of
    `valid` -- not the mean over the valid entries.  The reference's loss scale depends on it; kept."""
    assert valid.numel() > 0
    return torch.where(valid, t, t.new_zeros(())).mean() * (valid.float().sum() / valid.numel())


def gather_rows(t: torch.Tensor, index: torch.Tensor, valid: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
    """t[index] along dim 0 with `fill` where `valid` is False (invalid index values are not read: they are redirected to row 0 first);
    result shape index.shape + t.shape[1:] (cpc_aux_loss.py:34-61, dim 0)."""
    flat = torch.where(valid, index, torch.zeros_like(index)).reshape(-1)
    rows = t.index_select(0, flat)
    keep = valid.reshape(-1, *([1] * (t.dim() - 1)))
    return torch.where(keep, rows, rows.new_full((), fill)).reshape(*index.shape, *t.shape[1:])


class ActionConditionedForwardModelingLoss(nn.Module):
    """Base of the forward-modelling losses (cpc_aux_loss.py:64-224): the action embedding, the prediction LSTM and the index building."""

    def __init__(self, action_space, hidden_size: int, k: int = 20, time_subsample: int = 6, future_subsample: int = 2):
        super().__init__()
        self._action_embed = ActionEmbedding(action_space)
        self._future_predictor = nn.LSTM(self._action_embed.output_size, hidden_size)
        self.k, self.time_subsample, self.future_subsample = k, time_subsample, future_subsample
        self._hidden_size = hidden_size
        self.layer_init()

    def layer_init(self):
        """Orthogonal matrices (the embedding table included), zero biases (cpc_aux_loss.py:102-107)."""
        for name, p in self.named_parameters():
            if "weight" in name and p.dim() > 1:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    # the module's random draws, in one place (tests route them through the CPU generator to compare a device run with a CPU run)
    def _randperm(self, n: int, device) -> torch.Tensor: return torch.randperm(n, device=device, dtype=torch.int64)

    def _multinomial(self, probs: torch.Tensor, num_samples: int, replacement: bool) -> torch.Tensor:
        return torch.multinomial(probs, num_samples=num_samples, replacement=replacement)

    def _start_steps(self, lengths: torch.Tensor, device):
        """-> (start step, sequence index, last step of that sequence), one entry per sampled start, sequences in pack order.
        A sequence with len - 1 <= time_subsample contributes steps 1 .. len-1 (none for a length-1 fragment), a longer one the first
        `time_subsample` entries of randperm(len) -- step 0 and the last step included (cpc_aux_loss.py:124-150)."""
        starts, owner = [], []
        for s, n in enumerate(lengths.tolist()):
            if self.time_subsample >= n - 1:
                st = torch.arange(1, n, device=device, dtype=torch.int64)
            else:
                st = self._randperm(n, device)[: self.time_subsample]
            starts.append(st)
            owner.append(torch.full_like(st, s))
        starts, owner = torch.cat(starts), torch.cat(owner)
        return starts, owner, lengths.to(device)[owner] - 1

    def _build_inds(self, info):
        """-> action_inds, target_inds [k, M] (rows of the minibatch), action_valids, target_valids [k, M].
        Packed layout: step t of the pack holds num_seqs_at_step[t] rows, one per sequence still running, sequences ordered by
        decreasing length -- so (sequence s, step t) is packed row first_row[t] + s, and select_inds maps packed rows to minibatch rows.
        Step start+j is an action step while it is before the sequence's last step, and has a target while start+j+1 is too."""
        nseq = info["num_seqs_at_step"]
        device = nseq.device
        lengths = info["cpu_sequence_lengths"] if "cpu_sequence_lengths" in info else info["sequence_lengths"].cpu()
        start, owner, last = self._start_steps(lengths, device)
        step = torch.arange(self.k, device=device, dtype=torch.int64)[:, None] + start[None, :]
        action_valids = step < last[None, :]
        target_valids = step + 1 < last[None, :]
        step = torch.where(action_valids, step, torch.zeros_like(step))
        first_row = torch.cumsum(nseq, 0) - nseq
        select = info["select_inds"]
        action_inds = select[first_row[step] + owner[None, :]]
        target_inds = select[first_row[step + 1] + owner[None, :]]
        return action_inds, target_inds, action_valids, target_valids

    def _kept_predictions(self, k: int, M: int, like: torch.Tensor) -> torch.Tensor:
        """Flat indices into the [k, M]-flattened predictions that enter the loss (cpc_aux_loss.py:191-214).  `future_subsample` draws
        without replacement from 0..k-1 per start m, offset by m * k.  NOTE the offset is m * k on a tensor flattened with stride M per
        step: entry m*k + j is (step (m*k+j) // M, start (m*k+j) % M), not (step j, start m).  The same flat index is applied to the
        targets and validity masks, so every kept row is still a consistent (prediction, target) pair; it is the SELECTION that differs
        from the docstring's intent.  Kept: the reference's loss values are the contract."""
        if self.future_subsample < k:
            pick = self._multinomial(like.new_full((), 1.0 / k).expand(M, k), self.future_subsample, False)
        else:
            pick = torch.arange(k, device=like.device, dtype=torch.int64)[:, None].expand(k, M)
        base = torch.arange(0, pick.size(0) * k, k, device=pick.device, dtype=pick.dtype)[:, None]
        return (pick + base).flatten()

    def forward(self, aux_loss_state, batch):
        act = self._action_embed(batch["action"])
        action_inds, target_inds, action_valids, target_valids = self._build_inds(batch["rnn_build_seq_info"])
        h0 = gather_rows(aux_loss_state["rnn_output"], action_inds[0], action_valids[0]).unsqueeze(0)
        act = gather_rows(act, action_inds, action_valids)
        # k <= 20 steps of [M, 32 + H] x [.., 4H]: ATen's own recurrent cell + BLAS GEMMs; the vendor RNN library (MIOpen behind the
        # cudnn flag) would compile and tune kernels at first use for a tensor this small -- the package has no other dependency on it
        with torch.backends.cudnn.flags(enabled=False):
            preds, _ = self._future_predictor(act, (h0, h0))
        kept = self._kept_predictions(act.size(0), act.size(1), act)
        return preds, action_inds, target_inds, kept, action_valids, target_valids


@baseline_registry.register_auxiliary_loss(name="cpca")
class CPCA(ActionConditionedForwardModelingLoss):
    """cpc_aux_loss.py:227-355.  `net` supplies output_size (recurrent features) and perception_embedding_size (visual fc output);
    a blind net has no perception embedding and is refused, as in the reference."""

    def __init__(self, action_space, net, k: int = 20, time_subsample: int = 6, future_subsample: int = 2, num_negatives: int = 20,
                 loss_scale: float = 0.1):
        assert not net.is_blind, "CPCA only works for networks with a visual encoder"
        hidden, embed = net.output_size, net.perception_embedding_size
        super().__init__(action_space, hidden, k, time_subsample, future_subsample)
        # score(p, e) = head(relu(A p + a + B e)): the first layer of the head is split so that A p is computed once per kept
        # prediction and B e once per candidate embedding instead of once per pair
        self._predictor_first_layers = nn.ModuleList([nn.Linear(hidden, hidden, bias=True), nn.Linear(embed, hidden, bias=False)])
        self._predictor = nn.Sequential(nn.ReLU(True), nn.Linear(hidden, hidden), nn.ReLU(True), nn.Linear(hidden, 1))
        self.num_negatives, self.loss_scale = num_negatives, loss_scale
        self.layer_init()  # again, over ALL parameters (cpc_aux_loss.py:276): the base's matrices are re-drawn

    def forward(self, aux_loss_state, batch):
        preds, _, target_inds, kept, _, target_valids = super().forward(aux_loss_state, batch)
        embeds = aux_loss_state["perception_embed"]
        P = embeds.size(0)
        query = self._predictor_first_layers[0](preds.flatten(0, 1)[kept])  # [Q, H]
        pos_row = target_inds.flatten()[kept]
        valid = target_valids.flatten()[kept]
        Q = pos_row.size(0)

        pos = gather_rows(embeds, pos_row, valid)
        pos_logit = self._predictor(query + self._predictor_first_layers[1](pos))
        pos_loss = masked_mean(F.binary_cross_entropy_with_logits(pos_logit, torch.ones_like(pos_logit), reduction="none"),
                               valid.view(-1, 1))

        # negatives: uniform over the minibatch rows other than the positive's
        w = embeds.new_ones(Q, P)
        w[torch.arange(Q, device=w.device), pos_row] = 0.0
        w = w / w.sum(-1, keepdim=True)
        neg_row = self._multinomial(w, self.num_negatives, self.num_negatives > P)
        neg = embeds.index_select(0, neg_row.flatten()).view(Q, self.num_negatives, -1)
        neg_logit = self._predictor(query.unsqueeze(1) + self._predictor_first_layers[1](neg))
        neg_loss = masked_mean(F.binary_cross_entropy_with_logits(neg_logit, torch.zeros_like(neg_logit), reduction="none"),
                               valid.view(-1, 1, 1))
        return dict(loss=self.loss_scale * (pos_loss + neg_loss))
