"""Duck-typed stand-ins for the diffusers modules the TokenFlow hooks patch.

The reference hooks never import diffusers: they match classes by *name*
(/root/reference/util.py:46-58) and consume a handful of members
(SURVEY.md §8b).  These stand-ins carry exactly those members so that both the
verbatim reference (oracle/ref_loader.py) and the drop-in `tokenflow_utils`
module of this repo can be driven without diffusers, on CPU or GPU.

Class names matter: `BasicTransformerBlock` must be spelled exactly so
(`isinstance_str(module, "BasicTransformerBlock")`,
/root/reference/tokenflow_utils.py:10,16,204,286,439).
"""
import torch
import torch.nn as nn


class Attention(nn.Module):
    """Members used: to_q,to_k,to_v,to_out,heads,scale,head_to_batch_dim,
    batch_to_head_dim (/root/reference/tokenflow_utils.py:108-122,140-148,197)."""

    def __init__(self, query_dim, heads, cross_dim=None, out_bias=True):
        super().__init__()
        assert query_dim % heads == 0
        self.heads = heads
        self.scale = (query_dim // heads) ** -0.5
        cross_dim = query_dim if cross_dim is None else cross_dim
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(cross_dim, query_dim, bias=False)
        self.to_v = nn.Linear(cross_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def head_to_batch_dim(self, t):
        b, s, d = t.shape
        h = self.heads
        return t.reshape(b, s, h, d // h).permute(0, 2, 1, 3).reshape(b * h, s, d // h)

    def batch_to_head_dim(self, t):
        bh, s, dh = t.shape
        h = self.heads
        return t.reshape(bh // h, h, s, dh).permute(0, 2, 1, 3).reshape(bh // h, s, dh * h)

    def forward(self, x, encoder_hidden_states=None, attention_mask=None):
        ctx = x if encoder_hidden_states is None else encoder_hidden_states
        q = self.head_to_batch_dim(self.to_q(x))
        k = self.head_to_batch_dim(self.to_k(ctx))
        v = self.head_to_batch_dim(self.to_v(ctx))
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * self.scale, dim=-1)
        return self.to_out[0](self.batch_to_head_dim(torch.bmm(p, v)))


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, dim), nn.GELU(), nn.Linear(dim, dim))

    def forward(self, x):
        return self.net(x)


class BasicTransformerBlock(nn.Module):
    """Members used by TokenFlowBlock.forward
    (/root/reference/tokenflow_utils.py:300-427)."""

    def __init__(self, dim, heads, cross_dim=32):
        super().__init__()
        self.only_cross_attention = False
        self.use_ada_layer_norm = False
        self.use_ada_layer_norm_zero = False
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, timestep=None, cross_attention_kwargs=None,
                class_labels=None):
        h = hidden_states
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states) + h
        return self.ff(self.norm3(h)) + h


class AdaLayerNormZero(nn.Module):
    """`norm1` of an AdaLayerNormZero block as TokenFlowBlock.forward consumes it
    (/root/reference/tokenflow_utils.py:317-320): norm1(x, timestep, class_labels, hidden_dtype=) ->
    (x, gate_msa, shift_mlp, scale_mlp, gate_mlp), diffusers' contract, with a ONE-row conditioning embedding
    (the only batch shape for which the reference's 4-D hidden_states broadcast at all)."""

    def __init__(self, dim):
        super().__init__()
        self.emb = nn.Linear(1, dim)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(dim, 6 * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward(self, x, timestep, class_labels=None, hidden_dtype=None):
        emb = self.linear(self.silu(self.emb(timestep.reshape(-1, 1).float())))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        x = self.norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


class Transformer2DModel(nn.Module):
    def __init__(self, dim, heads, cross_dim=32):
        super().__init__()
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, cross_dim)])

    def forward(self, x, encoder_hidden_states=None):
        return self.transformer_blocks[0](x, encoder_hidden_states=encoder_hidden_states)


class ResnetBlock2D(nn.Module):
    """Members used by conv_forward (/root/reference/tokenflow_utils.py:51-98)."""

    def __init__(self, cin, cout, temb_ch=16, groups=4):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin)
        self.nonlinearity = nn.SiLU()
        self.upsample = None
        self.downsample = None
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.time_embedding_norm = "default"
        self.norm2 = nn.GroupNorm(groups, cout)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.output_scale_factor = 1.0

    def forward(self, input_tensor, temb):
        h = self.conv1(self.nonlinearity(self.norm1(input_tensor)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        x = input_tensor if self.conv_shortcut is None else self.conv_shortcut(input_tensor)
        return (x + h) / self.output_scale_factor


class _AttnStage(nn.Module):
    def __init__(self, dim, heads, n_attn, n_res=0, cross_dim=32):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(dim, heads, cross_dim) for _ in range(n_attn)])
        self.resnets = nn.ModuleList([ResnetBlock2D(dim, dim) for _ in range(n_res)])


class FakeUNet(nn.Module):
    """Skeleton with the exact topology the hooks index
    (/root/reference/tokenflow_utils.py:21-40,102,208-214):
    down_blocks[0..2].attentions[0..1], mid_block.attentions[0],
    up_blocks[1..3].attentions[0..2], up_blocks[1].resnets[1].
    `dims` = channel width per level (L0,L1,L2); mid uses L2's width."""

    def __init__(self, dims=(32, 64, 128), heads=4, cross_dim=32):
        super().__init__()
        d0, d1, d2 = dims
        self.down_blocks = nn.ModuleList([
            _AttnStage(d0, heads, 2), _AttnStage(d1, heads, 2), _AttnStage(d2, heads, 2),
            _AttnStage(d2, heads, 0)])
        self.mid_block = _AttnStage(d2, heads, 1)
        self.up_blocks = nn.ModuleList([
            _AttnStage(d2, heads, 0), _AttnStage(d2, heads, 3, n_res=3),
            _AttnStage(d1, heads, 3), _AttnStage(d0, heads, 3)])

    def transformer_blocks_in_order(self):
        """The 16 blocks in UNet execution order with their level id."""
        out = []
        for lvl in range(3):
            for a in self.down_blocks[lvl].attentions:
                out.append((lvl, a.transformer_blocks[0]))
        out.append((3, self.mid_block.attentions[0].transformer_blocks[0]))
        for res, lvl in ((1, 2), (2, 1), (3, 0)):
            for a in self.up_blocks[res].attentions:
                out.append((lvl, a.transformer_blocks[0]))
        return out


class FakePipeline(nn.Module):
    """Stands for the reference's `TokenFlow` wrapper: has `.unet`
    (/root/reference/run_tokenflow_pnp.py:25-68) plus extra non-UNet modules so
    that register_pivotal/register_batch_idx walk more than the UNet."""

    def __init__(self, **kw):
        super().__init__()
        self.unet = FakeUNet(**kw)
        self.text_encoder = nn.Linear(4, 4)


# ----------------------------------------------------------------------------------- a UNet the DRIVER can call
class RunnableUNet(FakeUNet):
    """`FakeUNet` with a forward pass, so that the reference's driver methods (`denoise_step`,
    /root/reference/run_tokenflow_pnp.py:195-217) can call `self.unet(latents, t, encoder_hidden_states=...)['sample']`.
    Same module tree as the hooks index; the data flow is a small U: conv_in -> 3 down levels (2 transformer blocks
    each, average-pool + 1x1 conv between levels) -> mid block -> 3 up levels (nearest upsample + 1x1 conv + skip;
    3 transformer blocks each, up_blocks[1] with a ResnetBlock2D in front of every block: resnets[1] is the one the
    feature injection patches) -> conv_out.  The extra layers are created AFTER the base class's, so a FakeUNet and
    a RunnableUNet built from the same seed share the hooks' 16 blocks."""

    def __init__(self, dims=(32, 64, 128), heads=4, cross_dim=32, latent_ch=4):
        super().__init__(dims=dims, heads=heads, cross_dim=cross_dim)
        d0, d1, d2 = dims
        self.conv_in = nn.Conv2d(latent_ch, d0, 3, padding=1)
        self.time_proj = nn.Linear(1, 16)
        self.down = nn.ModuleList([nn.Conv2d(d0, d1, 1), nn.Conv2d(d1, d2, 1), nn.Conv2d(d2, d2, 1)])
        self.up = nn.ModuleList([nn.Conv2d(d1, d0, 1), nn.Conv2d(d2, d1, 1), nn.Conv2d(d2, d2, 1)])
        self.conv_out = nn.Conv2d(d0, latent_ch, 3, padding=1)

    @staticmethod
    def _tokens(attn, h, enc):
        B, C, H, W = h.shape
        tok = attn(h.flatten(2).transpose(1, 2), encoder_hidden_states=enc)
        return tok.transpose(1, 2).reshape(B, C, H, W)

    def forward(self, sample, t, encoder_hidden_states=None):
        B = sample.shape[0]
        temb = self.time_proj(torch.as_tensor(t, dtype=torch.float32, device=sample.device).reshape(1, 1) / 1000.0)
        temb = temb.expand(B, -1)
        h = self.conv_in(sample)
        skips = []
        for lvl in range(3):
            for a in self.down_blocks[lvl].attentions:
                h = self._tokens(a, h, encoder_hidden_states)
            skips.append(h)
            h = self.down[lvl](nn.functional.avg_pool2d(h, 2))
        h = self._tokens(self.mid_block.attentions[0], h, encoder_hidden_states)
        for res, lvl in ((1, 2), (2, 1), (3, 0)):
            h = self.up[lvl](nn.functional.interpolate(h, scale_factor=2.0, mode="nearest")) + skips[lvl]
            stage = self.up_blocks[res]
            for i, a in enumerate(stage.attentions):
                if len(stage.resnets):
                    h = stage.resnets[i](h, temb)
                h = self._tokens(a, h, encoder_hidden_states)
        return {"sample": self.conv_out(h)}


class FakeDDIMScheduler:
    """What the driver uses of diffusers' DDIMScheduler: `timesteps` (a descending tensor) and
    `step(noise_pred, t, x)['prev_sample']` (/root/reference/run_tokenflow_pnp.py:216, 246) -- the deterministic
    (eta = 0) DDIM update on a 1000-step scaled-linear schedule."""

    def __init__(self, n_steps, first=951):
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.stride = 1000 // 20
        self.timesteps = torch.tensor([first - self.stride * i for i in range(n_steps)])

    def step(self, noise_pred, t, x):
        t = int(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[t - self.stride] if t - self.stride >= 0 else torch.tensor(1.0)
        x0 = (x - (1 - a_t) ** 0.5 * noise_pred) / a_t ** 0.5
        return {"prev_sample": a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * noise_pred}
