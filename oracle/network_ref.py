"""Oracle: TRU-Net body, restated with stock CPU torch.nn layers.  TEST INFRASTRUCTURE.

Follows the reference block classes (``/root/reference/network.py:9-120``) and the
canonical repaired composition R1-R4 of SURVEY.md section 0.2 for the broken
``TRUNet`` class (``network.py:122-171``).  Attribute names are the reference's so
that ``state_dict()`` carries the same 177 keys.

Layout everywhere: (N, C, L) with N = frames (batch), L = frequency positions.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _pw_bn_relu(cin, cout):
    return [nn.Conv1d(cin, cout, kernel_size=1), nn.BatchNorm1d(cout), nn.ReLU(inplace=True)]


class StandardConv1d(nn.Module):
    """network.py:9-21 -- Conv1d(k, s, padding=s//2) + ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.StandardConv1d = nn.Sequential(
            nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=stride // 2),
            nn.ReLU(inplace=True))

    def forward(self, x):
        return self.StandardConv1d(x)


class DepthwiseSeparableConv1d(nn.Module):
    """network.py:24-43 -- pw conv, BN, ReLU, depthwise conv(k, s, padding=k//2), BN, ReLU."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.DepthwiseSeparableConv1d = nn.Sequential(
            *_pw_bn_relu(in_channels, out_channels),
            nn.Conv1d(out_channels, out_channels, kernel_size, stride=stride,
                      padding=kernel_size // 2, groups=out_channels),
            nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.DepthwiseSeparableConv1d(x)


class GRUBlock(nn.Module):
    """network.py:45-58 -- batch_first GRU over dim 1, transpose, pw conv + BN + ReLU."""

    def __init__(self, in_channels, hidden_size, out_channels, bidirectional):
        super().__init__()
        self.GRU = nn.GRU(in_channels, hidden_size, batch_first=True, bidirectional=bidirectional)
        width = hidden_size * (2 if bidirectional else 1)
        self.conv = nn.Sequential(*_pw_bn_relu(width, out_channels))

    def forward(self, x):
        seq, _ = self.GRU(x)
        return self.conv(seq.transpose(1, 2))


def _trcnn_body(in_channels, out_channels, kernel_size, stride, tail=True):
    layers = _pw_bn_relu(in_channels, out_channels)
    layers.append(nn.ConvTranspose1d(out_channels, out_channels, kernel_size,
                                     stride=stride, padding=stride // 2))
    if tail:
        layers += [nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers)


def _fit_and_cat(x1, x2):
    """network.py:95-98 -- pad (negative = crop) x1 to x2's length, concat on channels."""
    d = x2.size(2) - x1.size(2)
    x1 = F.pad(x1, [d // 2, d - d // 2, 0, 0])
    return torch.cat((x1, x2), 1)


class FirstTrCNN(nn.Module):
    """network.py:60-76."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.FirstTrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride)

    def forward(self, x):
        return self.FirstTrCNN(x)


class TrCNN(nn.Module):
    """network.py:79-100."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.TrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride)

    def forward(self, x1, x2):
        return self.TrCNN(_fit_and_cat(x1, x2))


class LastTrCNN(nn.Module):
    """network.py:102-120 -- no BN/ReLU after the transposed conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.LastTrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride, tail=False)

    def forward(self, x1, x2):
        return self.LastTrCNN(_fit_and_cat(x1, x2))


class TRUNet(nn.Module):
    """Repaired composition R1-R4 of network.py:122-171.

    R1: layer sizes of network.py:134-150.  R2: first conv takes ``input_size``
    channels; the other ctor args are accepted and ignored as in the reference
    (D3).  R3: concat skips, single transpose around FGRU.  R4: TGRU is
    registered, never executed (``use_tgru`` is the "next" row of section 8f).
    """

    def __init__(self, input_size=3, channels_input=64, channels_output=3, channels_hidden=128,
                 kernel_sizes=(5, 3), strides=(2, 1), tr_channels_input=192):
        super().__init__()
        self.encoder = nn.ModuleList([
            StandardConv1d(input_size, 64, 5, 2),
            DepthwiseSeparableConv1d(64, 128, 3, 1),
            DepthwiseSeparableConv1d(128, 128, 5, 2),
            DepthwiseSeparableConv1d(128, 128, 3, 1),
            DepthwiseSeparableConv1d(128, 128, 5, 2),
            DepthwiseSeparableConv1d(128, 128, 3, 2)])
        self.decoder = nn.ModuleList([
            FirstTrCNN(64, 64, 3, 2),
            TrCNN(192, 64, 5, 2),
            TrCNN(192, 64, 3, 1),
            TrCNN(192, 64, 5, 2),
            TrCNN(192, 64, 3, 1),
            LastTrCNN(128, 8, 5, 2)])
        self.FGRU = GRUBlock(128, 64, 64, bidirectional=True)
        self.TGRU = GRUBlock(64, 128, 64, bidirectional=False)

    def forward(self, x, return_intermediates=False):
        skips, inter = [], {}
        for i, block in enumerate(self.encoder):
            x = block(x)
            skips.append(x)
            inter["enc%d" % i] = x
        skips = skips[::-1]
        x = self.FGRU(x.transpose(1, 2))
        inter["fgru"] = x
        x = self.decoder[0](x)
        inter["dec0"] = x
        for i in range(1, 6):
            x = self.decoder[i](x, skips[i])
            inter["dec%d" % i] = x
        return (x, inter) if return_intermediates else x

    def forward_tgru(self, x, T):
        """use_tgru (docs/net.jpg): x holds B utterances of T frames, (B*T, C_in, 257); the TGRU
        GRUBlock (network.py:150, :45-58 forward) runs over time for every (utterance, frequency position) between
        FGRU and the decoder.  Build-defined composition (the reference never calls TGRU, D6)."""
        skips = []
        for block in self.encoder:
            x = block(x)
            skips.append(x)
        skips = skips[::-1]
        x = self.FGRU(x.transpose(1, 2))                                   # (B*T, 64, 16)
        N = x.shape[0]
        B = N // T
        seq = x.reshape(B, T, 64, 16).permute(0, 3, 1, 2).reshape(B * 16, T, 64)
        y = self.TGRU(seq)                                                 # GRUBlock.forward: (B*16, 64, T)
        x = y.reshape(B, 16, 64, T).permute(0, 3, 2, 1).reshape(N, 64, 16)
        x = self.decoder[0](x)
        for i in range(1, 6):
            x = self.decoder[i](x, skips[i])
        return x

    def stream_step(self, x, h=None):
        """Stateful streaming with the TGRU block between FGRU and the decoder (docs/net.jpg): x is
        one frame per stream (S, C_in, 257); every (stream, frequency position) is a sequence of the unidirectional
        nn.GRU of network.py:150 advanced by one step; h (1, S*16, 128) is carried.  Build-defined (the reference never
        calls TGRU, D6): parity for this path is against this restatement only."""
        skips = []
        for block in self.encoder:
            x = block(x)
            skips.append(x)
        skips = skips[::-1]
        x = self.FGRU(x.transpose(1, 2))                        # (S, 64, 16)
        S = x.shape[0]
        seq = x.permute(0, 2, 1).reshape(S * 16, 1, 64)
        out, h = self.TGRU.GRU(seq, h)                          # (S*16, 1, 128)
        y = self.TGRU.conv(out.transpose(1, 2))                 # (S*16, 64, 1)
        x = y.reshape(S, 16, 64).permute(0, 2, 1).contiguous()  # (S, 64, 16)
        x = self.decoder[0](x)
        for i in range(1, 6):
            x = self.decoder[i](x, skips[i])
        return x, h
