"""Drop-in for the model-side names of the reference's model.py, running on the HIP kernels.

Kept names and positional signatures: ``weights``, ``nan_to_num``, ``Encoder``,
``Attention``, ``Decoder``, ``Seq2Seq``.  Documented divergences:
  * ``Encoder(n_feats=120)``: the reference hard-codes 120 MFCC features (model.py:37-38); the
    keyword makes the F=80 benchmark shape reachable while ``Encoder()`` is unchanged.
  * ``Seq2Seq.forward`` returns (T,B,V) log-probs from a CTC head ``nn.Linear(512, V)`` +
    log_softmax (the reference's Decoder prints a shape and returns None, model.py:117; the
    consumer contract is model.py:323).  The attention decoder is out of scope.
  * dropout (model.py:45,51 p=0.5; model.py:42 p=0.3) is applied only in train() mode like the
    reference; eval() is the parity mode.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from . import hipops

HID = 256


def weights(m):
    """Xavier-normal on nn.Linear weights, bias 0.1 (model.py:19-25).  Used via model.apply."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight.data)
        nn.init.constant_(m.bias.data, 0.1)


def nan_to_num(t, mynan=0.):
    """Non-finite entries -> mynan (model.py:27-32 does this recursively, element by element)."""
    if torch.all(torch.isfinite(t)):
        return t
    return torch.where(torch.isfinite(t), t, torch.as_tensor(mynan, dtype=t.dtype, device=t.device))


class Encoder(nn.Module):
    """InstanceNorm -> Linear(F,512)+leaky_relu(+dropout) -> 3-layer BLSTM(256) (model.py:34-56).

    Parameters carry the reference's state_dict names: input_layer.{weight,bias},
    blstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{0,1,2}{,_reverse}."""

    def __init__(self, n_feats=120):
        super().__init__()
        self.n_feats = n_feats
        self.inst_norm = nn.InstanceNorm2d(n_feats)   # parameter-free; kept for state_dict/API parity
        self.input_layer = nn.Linear(n_feats, 512)
        self.blstm = nn.LSTM(input_size=512, hidden_size=HID, num_layers=3, dropout=0.3,
                             bidirectional=True, batch_first=True)
        self.drop = nn.Dropout()
        self.dropout_seed = 0x5EED          # masks are functions of (seed, call counter, index)
        self._drop_calls = 0

    def _layer_params(self, l):
        out = []
        for sfx in ("", "_reverse"):
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                out.append(getattr(self.blstm, f"{k}_l{l}{sfx}"))
        return out

    def forward_time_major(self, x, mask):
        """x (B,F,T), mask (B,T) -> (T,B,512) and int32 lengths."""
        if not x.is_cuda:
            raise RuntimeError("policy_gradient_asr_amd.Encoder runs on the MI355X only (no CPU fallback)")
        lengths = mask.sum(dim=1).to(torch.int32).contiguous()   # stays on the device: no host sync
        training = self.training
        fuse = not training
        y = Fh.InstNormAffineFn.apply(x.float(), self.input_layer.weight, self.input_layer.bias, fuse)
        h = y
        if training:                                   # nn.Dropout(), model.py:45,51
            h = Fh.DropoutFn.apply(h, self.drop.p, self.dropout_seed, self._next_drop_offset())
        for l in range(3):
            h = Fh.blstm_layer(h, lengths, self._layer_params(l), dact_y=y if (l == 0 and fuse) else None)
            if training and l < 2:                     # nn.LSTM(dropout=0.3), model.py:42: outputs of layers 0,1
                h = Fh.DropoutFn.apply(h, self.blstm.dropout, self.dropout_seed, self._next_drop_offset())
        return h, lengths

    def _next_drop_offset(self):
        self._drop_calls += 1
        return self._drop_calls

    def forward(self, x, mask):
        h, _ = self.forward_time_major(x, mask)
        return h.transpose(0, 1)   # (B,T,512) like model.py:56


class Attention(nn.Module):
    """Signature only (model.py:58-94); the attention decoder is out of scope."""

    def __init__(self):
        super().__init__()

    def forward(self, dec_t, enc_out):
        raise NotImplementedError("attention decoder is out of scope of the CTC/policy-gradient path")


class Decoder(nn.Module):
    """Signature only (model.py:99-117): Decoder(alphabet_size, hidden_size)."""

    def __init__(self, alphabet_size, hidden_size):
        super().__init__()
        self.alphabet_size = alphabet_size
        self.hidden_size = hidden_size

    def forward(self, target_inputs, encoder_outputs, device=None):
        raise NotImplementedError("attention decoder is out of scope of the CTC/policy-gradient path")


class Seq2Seq(nn.Module):
    """Seq2Seq(alphabet_size).forward(x, t, fmask, device) (model.py:174-183) with a CTC head."""

    def __init__(self, alphabet_size, n_feats=120):
        super().__init__()
        self.encoder = Encoder(n_feats=n_feats)
        self.head = nn.Linear(2 * HID, alphabet_size)

    def logits(self, x, fmask):
        """(T,B,V) pre-softmax scores and int32 lengths -- what the fused loss consumes."""
        h, lengths = self.encoder.forward_time_major(x, fmask)
        return Fh.LinearFn.apply(h, self.head.weight, self.head.bias), lengths

    def forward(self, x, t, fmask, device=None):
        z, _ = self.logits(x, fmask)
        return Fh.LogSoftmaxFn.apply(z)
