"""`ldm.modules.encoders.NVS_modules` (reference NVS_modules.py:92-270): prompt encoder of the novel-view-synthesis task model
(configs/nvs_training_config.yaml, BASELINE configs[4]).

Differences to the single-reference encoder (Refill_modules.py):
  * `RelPosModel` (92-106): a small MLP (4 -> 512 -> SiLU -> 1024) turns the relative camera pose of the target view into ONE
    token embedding that overwrites position `len(special_tokens) + 1` of the token embeddings -- the slot of the last learned
    token, shifted by <start_of_text> (219-224); with `pos_strengthen` a second head (SiLU -> 1024 -> 1024) overwrites the LAST
    position of the encoder output (244-251);
  * `inputs` is either the prompt list or `[prompts, rel_pos [B, 1, 4]]` (185-190);
  * optional per-view tokens "<view_direct-{j}-{l}>" instead of the pose MLP (144-148, 166-170);
  * training-time classifier-free dropout: with probability `cfg_rate` a sample's token embeddings are replaced by those of the
    empty prompt (226-233), and its pose output by the encoder's own last position (246-249).
The pose MLP is a [B, 4] x [4, 512] x [512, 1024] product per batch -- host PyTorch glue next to the text tower, which runs on the
HIP kernels like the base class.
"""
import torch
import torch.nn as nn

from ldm.modules.encoders.Refill_modules import (AbstractEncoder, IdentityEncoder, PromptCLIPEmbedder as _Base,  # noqa: F401
                                                 expand_special_tokens, init_special_embeddings, tokenize)
from ldm.modules.encoders.multiview_Refill_modules import view_token_names

NVS_VIEW_INIT_TEXT = "overhead view, front view, side view, back view"


class RelPosModel(nn.Module):
    def __init__(self, input_ch=3, out_ch=1024, pos_strengthen=False):
        super().__init__()
        self.mlp1 = nn.Sequential(nn.Linear(input_ch, out_ch // 2), nn.SiLU(), nn.Linear(out_ch // 2, out_ch))
        self.pos_strengthen = pos_strengthen
        if pos_strengthen:
            self.mlp2 = nn.Sequential(nn.SiLU(), nn.Linear(out_ch, out_ch))

    def forward(self, x):
        first = self.mlp1(x)
        return (first, self.mlp2(first)) if self.pos_strengthen else first


class NVSCLIPEmbedder(_Base):
    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 special_tokens=("<left>", "<right>"), init_text=None, tokenwise_init=False, deep_prompt=False, cross_attn_layers=16,
                 view_prompt=False, view_num=None, view_token_len=1, pos_strengthen=False, cfg_rate=0.0, **kwargs):
        AbstractEncoder.__init__(self)
        names, texts = expand_special_tokens(special_tokens, init_text, deep_prompt, cross_attn_layers)
        if view_prompt:
            extra = view_token_names(view_num, view_token_len, closing=">")
            names = names + extra
            texts = (texts or []) + [NVS_VIEW_INIT_TEXT] * len(extra)
        self.cfg_rate = cfg_rate
        self.pos_strengthen = pos_strengthen
        self._build(arch, version, device, max_length, layer, names, texts, tokenwise_init, deep_prompt, cross_attn_layers)
        # registered after the tower like the reference (parameter order = optimizer / state-dict order)
        self.rel_pos_model = None if view_prompt else RelPosModel(input_ch=4, out_ch=1024, pos_strengthen=pos_strengthen)
        if freeze:
            self.freeze()

    def forward(self, inputs):
        if len(inputs) > 1 and isinstance(inputs[1], torch.Tensor):
            text, rel_pos = inputs
        else:
            text, rel_pos = inputs, None
        tokens, shape = self._tokens(text, self.deep_prompt)
        x = self._embed(tokens)
        pose_out = None
        if rel_pos is not None:
            pose = self.rel_pos_model(rel_pos)
            pose_in, pose_out = pose if self.pos_strengthen else (pose, None)
            # [B, 1, C] into the slot of the last learned token (shifted by <start_of_text>)
            x = x.clone()
            x[:, len(self.special_tokens) + 1, :] = pose_in.to(x.dtype).reshape(x.shape[0], -1)
        drop = None
        if self.cfg_rate > 0.0 and self.training:
            null = self.model.token_embedding(tokenize(self.tokenizer, [""]).to(x.device))       # [1, 77, C]
            drop = (torch.rand(x.shape[0]) < self.cfg_rate).to(dtype=torch.float32, device=x.device).reshape(-1, 1, 1)
            x = (1 - drop) * x + drop * null
        z = self.encode_with_transformer(x)
        if shape is not None:
            z = z.reshape(shape[0], shape[1], shape[2], -1)
        if pose_out is not None:
            pose_out = pose_out.to(x.dtype).reshape(z.shape[0], -1)
            if drop is not None:
                pose_out = pose_out * (1 - drop[:, 0]) + z[:, -1, :] * drop[:, 0]
            z = z.clone()
            z[:, -1, :] = pose_out
        return z
