"""Generator: GCN feature extractor, upsampling head, binary-mask head, position expansion.

Host-side mirror of the reference's `upsampling_network.py` (GCNFeatureExtractor :7-41,
UpsamplingModule :44-74, BinaryMaskingModule :77-104, SRNet :108-185, NoMaskSRNet :189-223)
with identical parameter names.  `forward_frames` additionally runs several frames of a
clip through the network body as ONE batch (the generator has no cross-sample coupling:
no norm layers), which is how the train step uses it on MI355X.
"""
import torch
import torch.nn as nn

from .graph_conv import (EdgeConv, IDGCNLayer, build_shared_mlp, conv_bn_layer, rows_first, rows_linear,
                         rows_seq)


# generator -> side stream for its mask head (set by gan_step_graph; a registry, not an attribute: modules are deep-copied
# by tests and tools, streams are not)
import weakref

_AUX_STREAMS = weakref.WeakKeyDictionary()


def set_aux_stream(net, stream):
    """Run `net`'s mask head on `stream`, beside its upsampling head (None: back to one stream)."""
    if stream is None:
        _AUX_STREAMS.pop(net, None)
    else:
        _AUX_STREAMS[net] = stream


class GCNFeatureExtractor(nn.Module):
    def __init__(self, layer_num, in_node_feat_dim, out_node_feat_dim, node_emb_dim=128):
        super().__init__()
        self.conv_layers = nn.ModuleList()
        for l in range(layer_num):
            if l == 0:
                self.conv_layers.append(EdgeConv(in_node_feat_dim, node_emb_dim, bn=False, insn=False,
                                                 k=20, mlp_layer=True))
            elif l == layer_num - 1:
                self.conv_layers.append(IDGCNLayer(node_emb_dim, out_node_feat_dim, bn=False,
                                                   insn=False, residual=True))
            else:
                self.conv_layers.append(IDGCNLayer(node_emb_dim, node_emb_dim, bn=False, insn=False,
                                                   ln=False, residual=True))

    def forward_rows(self, feature, pos=None):
        """feature (B,N,C) [, pos (B,N,3)] -> (B,N,C') rows (the build's internal layout)."""
        x = feature
        outs = []
        for l, layer in enumerate(self.conv_layers):
            if l == 0:
                x = layer.forward_rows(x, pos)
            else:
                x = layer.forward_rows(x)
                outs.append(x)
        return torch.cat(outs, dim=-1)

    def forward(self, feature, pos=None):
        """Reference signature: (B,N,C) -> (B,C',N,1)."""
        if rows_first():
            return self.forward_rows(feature, pos).transpose(1, 2).unsqueeze(-1)
        x = feature.permute(0, 2, 1).contiguous()                  # reference order, (B,C,N) planes
        outs = []
        for l, layer in enumerate(self.conv_layers):
            if l == 0:
                x = layer(x, pos) if pos is not None else layer(x)
            else:
                x = layer(x)
                outs.append(x)
        return torch.cat(outs, dim=1)


def _head_layers(width, make_last_edgeconv):
    # modules are created in forward order so that seeded initialisation matches the reference
    layers = nn.ModuleList()
    layers.append(conv_bn_layer(width, width // 4, norm="none"))
    layers.append(EdgeConv(width // 4, width, aggregate="max", mlp_layer=True, k=12, bn=False, insn=False))
    layers.append(conv_bn_layer(width, width // 4, norm="none"))
    layers.append(make_last_edgeconv())
    return layers


class UpsamplingModule(nn.Module):
    def __init__(self, in_node_feat_dim, upsample_ratio, gcn_layer=2):
        super().__init__()
        if gcn_layer != 2:
            raise NotImplementedError("the reference only ever builds gcn_layer=2")
        out_dim = 3 * upsample_ratio
        self.upsample_ratio = upsample_ratio
        w = in_node_feat_dim
        self.upsample_layers = _head_layers(
            w, lambda: EdgeConv(w // 4, w, aggregate="max", mlp_layer=True, k=4, bn=False, insn=False))
        self.decoder = nn.Sequential(
            build_shared_mlp([w, out_dim // 2, out_dim], norm="none"),
            nn.Conv2d(out_dim, out_dim, 1, 1, 0, bias=True))

    def forward_rows(self, x):
        """(B,N,C) rows -> (B,N,3r)."""
        for layer in self.upsample_layers:
            x = layer.forward_rows(x) if isinstance(layer, EdgeConv) else rows_seq(layer, x)
        return rows_linear(self.decoder[1], rows_seq(self.decoder[0], x))

    def forward(self, feature):
        """Reference signature: (B,C,N,1) -> (B,N,3r)."""
        if rows_first():
            return self.forward_rows(feature.squeeze(-1).transpose(1, 2))
        for layer in self.upsample_layers:
            feature = layer(feature)
        return self.decoder(feature).squeeze(-1).permute(0, 2, 1).contiguous()


class BinaryMaskingModule(nn.Module):
    def __init__(self, in_node_feat_dim, gcn_layer=2):
        super().__init__()
        if gcn_layer != 2:
            raise NotImplementedError("the reference only ever builds gcn_layer=2")
        w = in_node_feat_dim
        self.upsample_layers = _head_layers(
            w, lambda: EdgeConv(w // 4, w, aggregate="sum", mlp_layer=False, k=8, bn=False, insn=False))
        self.decoder = nn.Sequential(
            build_shared_mlp([w, w // 2, w // 4], norm="none"),
            nn.Conv2d(w // 4, 1, 1, 1, 0, bias=True))

    def forward_rows(self, x):
        """(B,N,C) rows -> (B,N,1)."""
        for layer in self.upsample_layers:
            x = layer.forward_rows(x) if isinstance(layer, EdgeConv) else rows_seq(layer, x)
        return rows_linear(self.decoder[1], rows_seq(self.decoder[0], x), 0.0)      # ReLU = slope 0, in the epilogue

    def forward(self, feature):
        """Reference signature: (B,C,N,1) -> (B,N,1)."""
        if rows_first():
            return self.forward_rows(feature.squeeze(-1).transpose(1, 2))
        for layer in self.upsample_layers:
            feature = layer(feature)
        return torch.relu(self.decoder(feature)).squeeze(-1).permute(0, 2, 1).contiguous()


class SRNet(nn.Module):
    """Upsampling generator with the learned binary mask (upsampling_network.py:108-185)."""

    def __init__(self, in_feats, node_emb_dim, upsample_ratio=8, feature_extractor_depth=3):
        super().__init__()
        self.in_feats = in_feats
        self.feature_extractor = GCNFeatureExtractor(feature_extractor_depth, in_feats, node_emb_dim)
        width = node_emb_dim * (feature_extractor_depth - 1)
        self.upsampling_block = UpsamplingModule(width, upsample_ratio)
        self.filter_block = BinaryMaskingModule(width)
        self.upsample_ratio = upsample_ratio
        self.epsilon = 0.01
        # per frame of the most recent forward/forward_frames: did hard masking pad with 999?
        self.last_pad_flags = []

    # -- network body: everything up to (offsets, mask); batch rows are independent ----------
    def body(self, feature, pos):
        if rows_first():
            enc = self.feature_extractor.forward_rows(feature, pos if self.in_feats > 3 else None)
            aux = _AUX_STREAMS.get(self)
            if aux is not None and enc.is_cuda:
                # The two heads (upsampling_network.py:44-74 and :77-104) read the same encoding and nothing of each
                # other: on two streams they are parallel branches of a captured step -- forward AND backward (autograd
                # runs a node's backward on its forward stream) -- on a chain of ~5 us kernels that cannot fill the
                # chip alone (round 3; set by gan_step_graph).
                main = torch.cuda.current_stream(enc.device)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    mask = self.filter_block.forward_rows(enc).float()
                enc.record_stream(aux)
                edge = self.upsampling_block.forward_rows(enc).float()
                main.wait_stream(aux)
                mask.record_stream(main)
                return edge, mask
            return self.upsampling_block.forward_rows(enc).float(), self.filter_block.forward_rows(enc).float()
        enc = self.feature_extractor(feature, pos) if self.in_feats > 3 else self.feature_extractor(feature)
        return self.upsampling_block(enc).float(), self.filter_block(enc).float()

    def expand_pos_with_masking(self, pos, upsample_edge, binary_mask, hard_masking=False):
        """upsampling_network.py:131-157.  Returns (unpadded_pos, padded_or_compressed_pos|None)."""
        B = pos.shape[0]
        r = self.upsample_ratio
        keep = binary_mask.detach().view(B, -1, 1) > self.epsilon
        edge = upsample_edge * keep.float()
        expanded = pos.repeat(1, 1, r).view(B, -1, 3) + edge.view(B, -1, 3)
        self._padded = False
        if not hard_masking:
            return expanded, None
        hard = keep.repeat(1, 1, r)
        hard[:, :, 0] = True                                       # slot 0 always survives
        counts = hard.sum(dim=(1, 2))
        hard = hard.view(B, -1)
        if B > 1 and bool(torch.any(counts != counts.max())):      # host decision, as upstream
            padded = expanded.clone()
            padded[~hard] = 999
            self._padded = True
            return expanded, padded
        return expanded, expanded[hard].view(B, -1, 3)

    def expand_pos_static(self, pos, upsample_edge, binary_mask):
        """Hard masking without a host decision (for hipGraph capture): always the padded form
        `where(hard, expanded, 999)`, plus a device flag `all_keep`.  When every slot survives
        (`all_keep`), this IS what `expand_pos_with_masking(hard_masking=True)` returns; otherwise
        the caller must not use the static path (dummy handling needs the host)."""
        B = pos.shape[0]
        r = self.upsample_ratio
        keep = binary_mask.detach().view(B, -1, 1) > self.epsilon
        edge = upsample_edge * keep.float()
        expanded = pos.repeat(1, 1, r).view(B, -1, 3) + edge.view(B, -1, 3)
        hard = keep.repeat(1, 1, r)
        hard[:, :, 0] = True
        hard = hard.view(B, -1, 1)
        padded = torch.where(hard, expanded, torch.full_like(expanded, 999.0))
        return expanded, padded, hard.all()

    def forward(self, feature, pos, hard_masking=False):
        edge, mask = self.body(feature, pos)
        out_pos, padded = self.expand_pos_with_masking(pos, edge, mask, hard_masking=hard_masking)
        self.last_pad_flags = [self._padded]
        return out_pos, mask, padded

    def forward_frames(self, features, positions, hard_masking=False):
        """Lists of T per-frame (B,N,C)/(B,N,3) tensors -> list of T (pos, mask, padded).

        The body runs once on the T*B stacked clouds; masking/padding decisions are taken
        per frame exactly as T separate `forward` calls would (upsampling_network.py:147)."""
        T, B = len(positions), positions[0].shape[0]
        edge, mask = self.body(torch.cat(features, 0), torch.cat(positions, 0))
        outs, flags = [], []
        for t in range(T):
            sl = slice(t * B, (t + 1) * B)
            p, padded = self.expand_pos_with_masking(positions[t], edge[sl], mask[sl], hard_masking)
            outs.append((p, mask[sl], padded))
            flags.append(self._padded)
        self.last_pad_flags = flags
        return outs

    def forward_with_context(self, feature, pos, previous_mask):
        """Rollout with a 25-frame running mask average (upsampling_network.py:159-174)."""
        edge, mask = self.body(feature, None if self.in_feats <= 3 else pos)
        mask = torch.where(mask < 0.6, torch.zeros_like(mask), mask)
        mask = torch.where(mask > 0.6, torch.full_like(mask, 0.6), mask)
        if len(previous_mask) >= 25:
            previous_mask = previous_mask[-24:]
        previous_mask.append(mask)
        mask = torch.mean(torch.cat(previous_mask, dim=0), dim=0)
        _, out = self.expand_pos_with_masking(pos, edge, mask, hard_masking=True)
        return out, previous_mask


class NoMaskSRNet(nn.Module):
    """Generator without the mask head (upsampling_network.py:189-223)."""

    def __init__(self, in_feats, node_emb_dim, upsample_ratio=8, feature_extractor_depth=3):
        super().__init__()
        self.feature_extractor = GCNFeatureExtractor(feature_extractor_depth, in_feats, node_emb_dim)
        self.upsampling_block = UpsamplingModule(node_emb_dim * (feature_extractor_depth - 1), upsample_ratio)
        self.upsample_ratio = upsample_ratio

    def body(self, feature):
        if rows_first():
            return self.upsampling_block.forward_rows(self.feature_extractor.forward_rows(feature)).float()
        return self.upsampling_block(self.feature_extractor(feature)).float()

    def expand_pos(self, pos, upsample_edge):
        B = pos.shape[0]
        return pos.repeat(1, 1, self.upsample_ratio).view(B, -1, 3) + upsample_edge.view(B, -1, 3)

    def forward(self, feature, pos):
        if feature.dim() == 2:
            feature = feature.unsqueeze(0)
        if pos.dim() == 2:
            pos = pos.unsqueeze(0)
        edge = self.body(feature)
        out = self.expand_pos(pos, edge)
        return out, edge.view(out.shape[0], -1, 3)

    def forward_frames(self, features, positions):
        T, B = len(positions), positions[0].shape[0]
        edge = self.body(torch.cat(features, 0))
        outs = []
        for t in range(T):
            e = edge[t * B:(t + 1) * B]
            p = self.expand_pos(positions[t], e)
            outs.append((p, e.view(B, -1, 3)))
        return outs
