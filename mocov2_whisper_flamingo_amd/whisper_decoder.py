"""Whisper decoder on the HIP kernels: causal self-attention, cross-attention to
the encoder output and the GELU MLP of every layer, with the checkpoint key
names of ``whisper_model.decoder.*`` (the reference constructs the decoder with
``WhisperModel.from_pretrained`` at models/av_net.py:22 but never calls it; the
arithmetic is HF ``WhisperDecoder`` / ``WhisperDecoderLayer``,
modeling_whisper.py:649-797 / 416-505).

Forward only.  ``forward`` decodes a whole token sequence in one pass (teacher
forcing / scoring); ``new_cache`` + ``decode_step`` decode incrementally with the
self-attention keys / values of earlier tokens kept in a cache (what HF's
``DynamicCache`` holds, hf:649-797 ``past_key_values``), the causal attention
kernel taking the new tokens as the LAST queries of the cached sequence.
``cross_kv`` exposes the per-layer encoder projections so they are computed
once per encoder output.
"""
import torch
import torch.nn as nn

from . import ops
from .whisper_encoder import _Attention


class _DecoderLayer(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.encoder_attn = _Attention(d)
        self.encoder_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, 4 * d)
        self.fc2 = nn.Linear(4 * d, d)
        self.final_layer_norm = nn.LayerNorm(d)


class WhisperDecoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        d = cfg.d_model
        self.embed_tokens = nn.Embedding(cfg.vocab_size, d)
        self.embed_positions = nn.Embedding(cfg.max_target_positions, d)
        self.layers = nn.ModuleList([_DecoderLayer(d) for _ in range(cfg.decoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        self._packed = None
        self._proj = None

    # ---- weight packing -------------------------------------------------
    def _versions(self, dtype):
        return (dtype, sum(p._version for p in self.parameters()), next(self.parameters()).data_ptr())

    def _pack(self, dtype):
        """Per weight version: q/k/v of the self-attention fused into one [3D, D] matrix, k/v of the
        cross-attention into one [2D, D] matrix (k_proj has no bias), the head_dim**-0.5 query scaling folded
        into W_q and b_q (0.125: exact)."""
        key = self._versions(dtype)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        d = self.cfg.d_model
        if d // self.cfg.decoder_attention_heads != 64:
            raise ValueError("attention kernels are built for head_dim 64")
        s = 64 ** -0.5
        f32 = lambda t: t.detach().float().contiguous()
        cd = lambda t: t.detach().to(dtype).contiguous()
        ln = lambda m: (f32(m.weight), f32(m.bias))
        pk = {"tok": f32(self.embed_tokens.weight), "pos": f32(self.embed_positions.weight),
              "lnf": ln(self.layer_norm), "layers": []}
        for L in self.layers:
            a, x = L.self_attn, L.encoder_attn
            zb = torch.zeros_like(a.q_proj.bias)
            pk["layers"].append({
                "ln1": ln(L.self_attn_layer_norm),
                "wqkv": cd(torch.cat([a.q_proj.weight.detach() * s, a.k_proj.weight.detach(), a.v_proj.weight.detach()], 0)),
                "bqkv": f32(torch.cat([a.q_proj.bias.detach() * s, zb, a.v_proj.bias.detach()], 0)),
                "wo": cd(a.out_proj.weight), "bo": f32(a.out_proj.bias),
                "lnx": ln(L.encoder_attn_layer_norm),
                "wq_x": cd(x.q_proj.weight.detach() * s), "bq_x": f32(x.q_proj.bias.detach() * s),
                "wkv_x": cd(torch.cat([x.k_proj.weight.detach(), x.v_proj.weight.detach()], 0)),
                "bkv_x": f32(torch.cat([zb, x.v_proj.bias.detach()], 0)),
                "wo_x": cd(x.out_proj.weight), "bo_x": f32(x.out_proj.bias),
                "ln2": ln(L.final_layer_norm),
                "wfc1": cd(L.fc1.weight), "bfc1": f32(L.fc1.bias),
                "wfc2": cd(L.fc2.weight), "bfc2": f32(L.fc2.bias),
            })
        self._packed = (key, pk)
        return pk

    # ---- forward ----------------------------------------------------------
    @torch.no_grad()
    def cross_kv(self, encoder_hidden_states, dtype=torch.bfloat16):
        """Per-layer [B, Te, 2D] key|value projections of the encoder output (k_proj / v_proj of
        ``encoder_attn``): what HF keeps in the cross-attention half of its cache."""
        pk = self._pack(dtype)
        B, Te, D = encoder_hidden_states.shape
        e = encoder_hidden_states
        e = (e if (e.dtype == dtype and e.is_contiguous()) else ops.cast(e.contiguous(), dtype)).view(B * Te, D)
        return [ops.gemm_nt(e, lw["wkv_x"], bias=lw["bkv_x"]).view(B, Te, 2 * D) for lw in pk["layers"]]

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, encoder_hidden_states=None, dtype=torch.bfloat16, cross_kv=None):
        """input_ids int64 [B, T] (T <= max_target_positions); ``attention_mask`` [B, T] with 0 on padding
        tokens (keys of the self-attention only, as HF builds its causal mask); ``encoder_hidden_states``
        [B, Te, D] or ``cross_kv`` from :meth:`cross_kv`; with neither the cross-attention is skipped
        (hf:474).  Returns ``([B, T, D] in the compute dtype,)`` = ``last_hidden_state``."""
        cfg = self.cfg
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [batch, tokens]")
        B, T = input_ids.shape
        if T > cfg.max_target_positions:
            raise ValueError(f"{T} decoder tokens exceed max_target_positions={cfg.max_target_positions}")
        pk = self._pack(dtype)
        H, D = cfg.decoder_attention_heads, cfg.d_model
        use_mfma = dtype == torch.bfloat16
        xs = ops.embed_tokens(input_ids, pk["tok"], pk["pos"])           # fp32 residual stream [B*T, D]
        km = None if attention_mask is None else (attention_mask == 0).to(torch.uint8).contiguous()
        if cross_kv is None and encoder_hidden_states is not None:
            cross_kv = self.cross_kv(encoder_hidden_states, dtype)
        for i, lw in enumerate(pk["layers"]):
            h = ops.layernorm_fwd(xs, *lw["ln1"], out_dtype=dtype)
            qkv = ops.gemm_nt(h, lw["wqkv"], bias=lw["bqkv"]).view(B, T, 3 * D)
            o = ops.attn_fwd(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, key_mask=km, scale=1.0,
                             use_mfma=use_mfma, causal=True)
            ops.gemm_nt(o.view(B * T, D), lw["wo"], bias=lw["bo"], residual=xs, out=xs)
            if cross_kv is not None:
                kv = cross_kv[i]
                h = ops.layernorm_fwd(xs, *lw["lnx"], out_dtype=dtype)
                q = ops.gemm_nt(h, lw["wq_x"], bias=lw["bq_x"]).view(B, T, D)
                o = ops.attn_fwd(q, kv[:, :, :D], kv[:, :, D:], H, scale=1.0, use_mfma=use_mfma)
                ops.gemm_nt(o.view(B * T, D), lw["wo_x"], bias=lw["bo_x"], residual=xs, out=xs)
            h = ops.layernorm_fwd(xs, *lw["ln2"], out_dtype=dtype)
            f = ops.gemm_nt(h, lw["wfc1"], bias=lw["bfc1"], act=ops.ACT_GELU)
            ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], residual=xs, out=xs)
        out = ops.layernorm_fwd(xs, *pk["lnf"], out_dtype=dtype)
        return (out.view(B, T, D),)

    @torch.no_grad()
    def new_cache(self, batch, encoder_hidden_states=None, dtype=torch.bfloat16, max_tokens=None, device=None):
        """An empty decoding state for ``batch`` sequences: per-layer [B, max_tokens, 2D] key|value buffers
        (filled front to back by :meth:`decode_step`) and the encoder projections."""
        cfg = self.cfg
        n = cfg.max_target_positions if max_tokens is None else int(max_tokens)
        if n > cfg.max_target_positions:
            raise ValueError(f"max_tokens exceeds max_target_positions={cfg.max_target_positions}")
        device = device or self.embed_tokens.weight.device
        return {"len": 0, "dtype": dtype,
                "kv": [torch.empty((batch, n, 2 * cfg.d_model), dtype=dtype, device=device) for _ in self.layers],
                "cross": None if encoder_hidden_states is None else self.cross_kv(encoder_hidden_states, dtype)}

    @torch.no_grad()
    def decode_step(self, input_ids, cache):
        """Appends ``input_ids`` [B, t] (one token per call in greedy decoding, or a whole prompt) to the
        sequences in ``cache`` and returns their hidden states [B, t, D]: the same values ``forward`` gives
        for these positions of the full sequence (no padding mask: every cached token is attended)."""
        cfg = self.cfg
        dtype = cache["dtype"]
        B, T = input_ids.shape
        past = cache["len"]
        if past + T > cache["kv"][0].shape[1]:
            raise ValueError("decoding past the cache's max_tokens")
        pk = self._pack(dtype)
        H, D = cfg.decoder_attention_heads, cfg.d_model
        use_mfma = dtype == torch.bfloat16
        xs = ops.embed_tokens(input_ids, pk["tok"], pk["pos"], past=past)
        for i, lw in enumerate(pk["layers"]):
            h = ops.layernorm_fwd(xs, *lw["ln1"], out_dtype=dtype)
            qkv = ops.gemm_nt(h, lw["wqkv"], bias=lw["bqkv"]).view(B, T, 3 * D)
            kv = cache["kv"][i]
            for b in range(B):                                  # this step's keys | values behind the cached ones
                ops.cast_rows(qkv[b, :, D:], kv[b, past:past + T])
            seen = kv[:, :past + T]
            if B > 1 and past + T < kv.shape[1]:                # the kernel's batch stride is Tk rows: pack the filled part
                seen = seen.contiguous()
            o = ops.attn_fwd(qkv[:, :, :D], seen[:, :, :D], seen[:, :, D:], H, scale=1.0, use_mfma=use_mfma, causal=True)
            ops.gemm_nt(o.view(B * T, D), lw["wo"], bias=lw["bo"], residual=xs, out=xs)
            if cache["cross"] is not None:
                ckv = cache["cross"][i]
                h = ops.layernorm_fwd(xs, *lw["lnx"], out_dtype=dtype)
                q = ops.gemm_nt(h, lw["wq_x"], bias=lw["bq_x"]).view(B, T, D)
                o = ops.attn_fwd(q, ckv[:, :, :D], ckv[:, :, D:], H, scale=1.0, use_mfma=use_mfma)
                ops.gemm_nt(o.view(B * T, D), lw["wo_x"], bias=lw["bo_x"], residual=xs, out=xs)
            h = ops.layernorm_fwd(xs, *lw["ln2"], out_dtype=dtype)
            f = ops.gemm_nt(h, lw["wfc1"], bias=lw["bfc1"], act=ops.ACT_GELU)
            ops.gemm_nt(f, lw["wfc2"], bias=lw["bfc2"], residual=xs, out=xs)
        cache["len"] = past + T
        return ops.layernorm_fwd(xs, *pk["lnf"], out_dtype=dtype).view(B, T, D)

    @torch.no_grad()
    def greedy(self, prompt_ids, encoder_hidden_states, max_new_tokens, eos_token_id=None, dtype=torch.bfloat16):
        """Greedy decoding with the tied output projection: returns int64 [B, prompt + generated] (stops early
        when every sequence has produced ``eos_token_id``)."""
        B, P = prompt_ids.shape
        cache = self.new_cache(B, encoder_hidden_states, dtype, max_tokens=min(self.cfg.max_target_positions, P + max_new_tokens))
        out = [prompt_ids]
        h = self.decode_step(prompt_ids, cache)
        done = torch.zeros(B, dtype=torch.bool, device=prompt_ids.device)
        for _ in range(max_new_tokens):
            logits = self.project_tokens(h[:, -1:, :].contiguous())
            nxt = ops.argmax_rows(logits.view(B, -1)).view(B, 1)
            out.append(nxt)
            if eos_token_id is not None:
                done |= nxt.view(B) == eos_token_id
                if bool(done.all()):
                    break
            if cache["len"] >= cache["kv"][0].shape[1]:
                break
            h = self.decode_step(nxt, cache)
        return torch.cat(out, dim=1)

    @torch.no_grad()
    def project_tokens(self, hidden, out_dtype=torch.float32):
        """Vocabulary logits with the tied output projection of HF ``WhisperForConditionalGeneration``
        (``proj_out.weight`` is ``embed_tokens.weight``, no bias): [B, T, D] -> [B, T, vocab]."""
        key = (hidden.dtype, self.embed_tokens.weight._version, self.embed_tokens.weight.data_ptr())
        if self._proj is None or self._proj[0] != key:
            self._proj = (key, self.embed_tokens.weight.detach().to(hidden.dtype).contiguous())
        B, T, D = hidden.shape
        return ops.gemm_nt(hidden.reshape(B * T, D), self._proj[1], out_dtype=out_dtype).view(B, T, -1)
