"""VQAModel with the reference's constructor and forward signature (model/model.py:6-71).

Encoders are passed as classes, exactly as the reference does (`vision(out_features=512)`,
`text(hidden_dim=512)`, each returning `(seq (B,N,512), global (B,512))`); they are outside
the fusion hot path.  The fusion module runs on the HIP kernels; the classifier is the
Sequential of model/model.py:54-63 (the DP-SGD target, see vqa_amd.dp).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .fusion import Attention, CoAttention, Concat


class VQAModel(nn.Module):
    def __init__(self, vision, text, fusion_type="concat", hidden_dim=1024, num_classes=13,
                 use_token_pruning=False, prune_ratio=0.5, mixup_alpha=0.05, adversarial_mix=True,
                 adv_ratio=0.2, noise_injection=False, noise_scale=0.01, noise_type='gaussian'):
        super().__init__()
        self.fusion_type = fusion_type
        self.vision_encoder = vision(out_features=512)
        self.text_encoder = text(hidden_dim=512)
        if fusion_type == "concat":
            self.fusion_module = Concat(v_dim=self.vision_encoder.output_dim,
                                        q_dim=self.text_encoder.output_dim)
        elif fusion_type == "attention":
            # noise arguments are not forwarded here, as in model/model.py:26-34
            self.fusion_module = Attention(embed_dim=512, num_heads=8, use_pruning=use_token_pruning,
                                           prune_ratio=prune_ratio, mixup_alpha=mixup_alpha,
                                           adversarial_mix=adversarial_mix, adv_ratio=adv_ratio)
        elif fusion_type == "co_attention":
            self.fusion_module = CoAttention(embed_dim=512, num_heads=8, use_pruning=use_token_pruning,
                                             prune_ratio=prune_ratio, mixup_alpha=mixup_alpha,
                                             adversarial_mix=adversarial_mix, adv_ratio=adv_ratio,
                                             noise_injection=noise_injection, noise_scale=noise_scale,
                                             noise_type=noise_type)
        else:
            raise ValueError(f"Unknown fusion type: {fusion_type}")
        d = self.fusion_module.output_dim
        self.classifier = nn.Sequential(
            nn.Linear(d, hidden_dim * 2), nn.BatchNorm1d(hidden_dim * 2), nn.ReLU(), nn.Dropout(0.3),
            nn.Linear(hidden_dim * 2, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, num_classes))

    # Run the two encoders on two HIP streams: they are independent until the fusion, and the text
    # encoder's kernels (30 tokens per sample) are far too small to fill the GPU, so they fit into what
    # the vision encoder's leave idle.  autograd replays each node on the stream its forward ran on, so the
    # two backward passes overlap the same way.  Results are those of the sequential order.
    overlap_encoders = True

    def forward(self, images, input_ids, attention_mask):
        side = self._side_stream(images.device) if (self.overlap_encoders and images.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream(images.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                q_seq, q_global = self.text_encoder(input_ids, attention_mask)
            v_seq, v_global = self.vision_encoder(images)
            main.wait_stream(side)
            for t in (q_seq, q_global):
                t.record_stream(main)  # allocated on the side stream, consumed on this one
        else:
            v_seq, v_global = self.vision_encoder(images)
            q_seq, q_global = self.text_encoder(input_ids, attention_mask)
        fused = self.fusion_module(v_seq, v_global, q_seq, q_global)
        return self._classify(fused)

    def _classify(self, fused):
        """The classifier head.  Left as built (non-private training, bf16, on the GPU) its Linear layers run
        on the library's bf16 GEMM kernels like the encoders' do; wrapped for DP-SGD (vqa_amd.dp) or in any
        other precision it is simply called."""
        clf = self.classifier
        if not (type(clf) is nn.Sequential and fused.is_cuda and fused.dtype == torch.bfloat16):
            return clf(fused)
        from .encoders import _linear
        x = fused
        for layer in clf:
            x = _linear(x, layer) if type(layer) is nn.Linear else layer(x)
        return x

    @staticmethod
    def _side_stream(device):
        """The second stream of `device`, shared by every model (kept outside the module so that copying or
        pickling a model never meets a stream object)."""
        st = _SIDE_STREAMS.get(device.index)
        if st is None:
            if torch.cuda.is_current_stream_capturing():
                return None  # a stream cannot be created while a graph is captured: sequential order this time
            st = torch.cuda.Stream(device=device)
            _SIDE_STREAMS[device.index] = st
        return st


_SIDE_STREAMS: dict = {}


def load_weights(model: nn.Module, state_dict: dict) -> nn.Module:
    """Loads a reference checkpoint: DP checkpoints carry `classifier._module.*` keys, renamed to
    `classifier.*`; strict=False because DP models have GroupNorm where non-DP ones have
    BatchNorm running statistics (utils/util.py:113-128)."""
    renamed = {k.replace("classifier._module.", "classifier."): v for k, v in state_dict.items()}
    model.load_state_dict(renamed, strict=False)
    return model


def get_model_registry():
    """The three name -> class tables of the reference (utils/util.py:47-68), filled with this package's
    classes: what a configuration file's `Vision:`, `Text:` and `model:` entries resolve to.  `ViT` is an
    addition (the reference has no plain ViT wrapper); `VQAModel_IB`, the information-bottleneck variant that
    the reference excludes from DP-SGD training (train.py:177-178), is not built."""
    from . import encoders, vision
    vision_models = dict(vision.get_vision_registry(), ViT=encoders.ViT)
    text_models = {"Bert": encoders.Bert, "RoBerta": encoders.RoBerta, "BertQLoRA": encoders.BertQLoRA,
                   "RoBertaQLoRA": encoders.RoBertaQLoRA}
    return vision_models, text_models, {"VQAModel": VQAModel}
