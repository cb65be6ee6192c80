"""Generate the golden fixtures under tests/golden/ (run in the build container, where /root/reference and
HF transformers are available):

    python tests/golden/make_golden.py

  prompt_golden.npz     outputs of pixray's OWN Prompt / spherical_dist_loss (pixray.py:249-280), extracted from
                        /root/reference by AST and executed
  vq_clamp_golden.npz   outputs of pixray's OWN vector_quantize / ClampWithGrad (vqgan.py:48-79), same way
  clip_vit_golden.npz   an independent implementation of OpenAI's VisionTransformer (HF CLIPVisionModelWithProjection,
                        hidden_act=quick_gelu) on the seeded weights of pixray_amd.weights
  decoder_golden.npz    an independent implementation of taming's Decoder (HF JanusVQVAEDecoder, derived from taming)
  encoder_golden.npz    the same for taming's Encoder (HF JanusVQVAEEncoder)
  clip_text_golden.npz  an independent implementation of OpenAI's text tower (HF CLIPTextModelWithProjection)
                        on the seeded weights of pixray_amd.weights
  styleloss_golden.npz  outputs of pixray's OWN STROTSS functions and Vgg16_Extractor class (Losses/StyleLoss.py),
                        extracted by AST and executed on a torchvision-shaped VGG16 filled with the seeded weights of
                        pixray_amd.weights (torchvision and the pretrained net are absent): loss and d(loss)/d(image)
Weights are NOT stored: they are re-derived from the seeds by pixray_amd.weights.synthetic_*.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import _refextract as rx  # noqa: E402
from pixray_amd import weights  # noqa: E402

GOLDEN_CLIP = weights.ClipVitConfig("golden-tiny", 64, 16, 256, 2, 4, 64)
GOLDEN_VQ = weights.VqganConfig(ch=128, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=16,
                                z_channels=128, embed_dim=128, n_embed=256)


def hf_clip_from_params(cfg, p):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    hc = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=4 * cfg.width, num_hidden_layers=cfg.layers,
                          num_attention_heads=cfg.heads, image_size=cfg.input_resolution, patch_size=cfg.patch_size,
                          projection_dim=cfg.output_dim, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                          attention_dropout=0.0)
    m = CLIPVisionModelWithProjection(hc).eval()
    sd = {}
    w = cfg.width
    sd["vision_model.embeddings.class_embedding"] = p["class_embedding"]
    sd["vision_model.embeddings.patch_embedding.weight"] = p["conv1.weight"]
    sd["vision_model.embeddings.position_embedding.weight"] = p["positional_embedding"]
    sd["vision_model.pre_layrnorm.weight"] = p["ln_pre.weight"]
    sd["vision_model.pre_layrnorm.bias"] = p["ln_pre.bias"]
    for i in range(cfg.layers):
        a, b = f"transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
        wq, wk, wv = p[a + "attn.in_proj_weight"].split(w, 0)
        bq, bk, bv = p[a + "attn.in_proj_bias"].split(w, 0)
        for n, (ww, bb) in zip("qkv", ((wq, bq), (wk, bk), (wv, bv))):
            sd[b + f"self_attn.{n}_proj.weight"] = ww
            sd[b + f"self_attn.{n}_proj.bias"] = bb
        sd[b + "self_attn.out_proj.weight"] = p[a + "attn.out_proj.weight"]
        sd[b + "self_attn.out_proj.bias"] = p[a + "attn.out_proj.bias"]
        sd[b + "layer_norm1.weight"] = p[a + "ln_1.weight"]; sd[b + "layer_norm1.bias"] = p[a + "ln_1.bias"]
        sd[b + "layer_norm2.weight"] = p[a + "ln_2.weight"]; sd[b + "layer_norm2.bias"] = p[a + "ln_2.bias"]
        sd[b + "mlp.fc1.weight"] = p[a + "mlp.c_fc.weight"]; sd[b + "mlp.fc1.bias"] = p[a + "mlp.c_fc.bias"]
        sd[b + "mlp.fc2.weight"] = p[a + "mlp.c_proj.weight"]; sd[b + "mlp.fc2.bias"] = p[a + "mlp.c_proj.bias"]
    sd["vision_model.post_layernorm.weight"] = p["ln_post.weight"]
    sd["vision_model.post_layernorm.bias"] = p["ln_post.bias"]
    sd["visual_projection.weight"] = p["proj"].T.contiguous()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def hf_decoder_from_params(cfg, p):
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from transformers.models.janus.modeling_janus import JanusVQVAEDecoder
    jc = JanusVQVAEConfig(base_channels=cfg.ch, channel_multiplier=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                          latent_channels=cfg.z_channels, out_channels=cfg.out_ch, dropout=0.0)
    d = JanusVQVAEDecoder(jc).eval()
    nres = len(cfg.ch_mult)
    sd = {}
    for k, v in p.items():
        if not k.startswith("decoder."):
            continue
        k2 = k[len("decoder."):]
        if k2.startswith("up."):
            parts = k2.split(".")
            parts[1] = str(nres - 1 - int(parts[1]))      # Janus appends levels in forward order
            k2 = ".".join(parts)
        sd[k2] = v
    d.load_state_dict(sd, strict=True)
    return d


GOLDEN_TEXT = weights.ClipTextConfig("golden-text", vocab_size=300, context_length=24, width=256, layers=2, heads=4, output_dim=64)


def golden_tokens(cfg, n, g):
    """clip.tokenize-shaped ids: SOT (vocab-2), words, EOT (vocab-1, the largest id), zero padding"""
    tk = torch.zeros(n, cfg.context_length, dtype=torch.long)
    for i in range(n):
        L = int(torch.randint(1, cfg.context_length - 2, (1,), generator=g))
        tk[i, 0] = cfg.vocab_size - 2
        tk[i, 1:1 + L] = torch.randint(1, cfg.vocab_size - 2, (L,), generator=g)
        tk[i, 1 + L] = cfg.vocab_size - 1
    return tk


def hf_clip_text_from_params(cfg, p):
    """an independent implementation of OpenAI's text tower: HF CLIPTextModelWithProjection (eos_token_id=2 selects the
    OpenAI pooling rule `argmax(input_ids)`)"""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    hc = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.width, intermediate_size=4 * cfg.width,
                        projection_dim=cfg.output_dim, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                        max_position_embeddings=cfg.context_length, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                        eos_token_id=2, bos_token_id=0, pad_token_id=1)
    m = CLIPTextModelWithProjection(hc).eval()
    w = cfg.width
    sd = {"text_model.embeddings.token_embedding.weight": p["token_embedding.weight"],
          "text_model.embeddings.position_embedding.weight": p["positional_embedding"],
          "text_model.final_layer_norm.weight": p["ln_final.weight"], "text_model.final_layer_norm.bias": p["ln_final.bias"],
          "text_projection.weight": p["text_projection"].T.contiguous()}
    for i in range(cfg.layers):
        a, b = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
        for j, n_ in enumerate("qkv"):
            sd[b + f"self_attn.{n_}_proj.weight"] = p[a + "attn.in_proj_weight"][j * w:(j + 1) * w]
            sd[b + f"self_attn.{n_}_proj.bias"] = p[a + "attn.in_proj_bias"][j * w:(j + 1) * w]
        sd[b + "self_attn.out_proj.weight"] = p[a + "attn.out_proj.weight"]; sd[b + "self_attn.out_proj.bias"] = p[a + "attn.out_proj.bias"]
        for (x, y) in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
            sd[b + y + ".weight"] = p[a + x + ".weight"]; sd[b + y + ".bias"] = p[a + x + ".bias"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    return m


def hf_encoder_from_params(cfg, p):
    """an independent implementation of taming's Encoder: HF JanusVQVAEEncoder (LlamaGen VQGAN, derived from taming;
    same graph when attention sits only at the lowest resolution, as it does for imagenet_f16_16384)"""
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from transformers.models.janus.modeling_janus import JanusVQVAEEncoder
    jc = JanusVQVAEConfig(base_channels=cfg.ch, channel_multiplier=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                          latent_channels=cfg.z_channels, in_channels=3, double_latent=False, dropout=0.0)
    e = JanusVQVAEEncoder(jc).eval()
    sd = {k[len("encoder."):]: v for k, v in p.items() if k.startswith("encoder.")}
    e.load_state_dict(sd, strict=True)
    return e


def styleloss_golden(out_dir):
    ns = rx.styleloss_ns()
    params = weights.synthetic_vgg16_params(0)
    ex = rx.reference_vgg_extractor(ns, params)
    rec = {}
    for tag, (H, W, cw, seed) in {"a": (80, 72, 32.0, 123), "b": (70, 90, 16.0, 7)}.items():
        g = torch.Generator().manual_seed(seed)
        img = torch.rand(1, 3, H, W, generator=g)
        style = torch.rand(1, 3, H, W, generator=g)
        x = img.clone().requires_grad_(True)
        np.random.seed(seed)
        loss = ns["strotss_loss"](x, style, cw, extractor=ex)
        (gr,) = torch.autograd.grad(loss, x)
        rec.update({f"img_{tag}": img.numpy(), f"style_{tag}": style.numpy(), f"cw_{tag}": np.float32(cw), f"seed_{tag}": np.int64(seed),
                    f"loss_{tag}": loss.detach().numpy(), f"grad_{tag}": gr.numpy()})
    np.random.seed(5)
    rec["hyper"] = ex.forward_samples_hypercolumn(torch.from_numpy(rec["style_a"]), samps=40).numpy()
    np.savez_compressed(os.path.join(out_dir, "styleloss_golden.npz"), **rec)


def main():
    out = HERE
    styleloss_golden(out)
    # ---- pixray's own fragments ----------------------------------------------------------------------------------
    ns = rx.pixray_prompt_ns()
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 64, generator=g)
    e = torch.randn(3, 64, generator=g)
    rec = {"x": x.numpy(), "embed": e.numpy()}
    for tag, (w, stop) in {"a": (1.0, float("-inf")), "b": (-0.7, float("-inf")), "c": (2.0, 1.2)}.items():
        xr = x.clone().requires_grad_(True)
        loss = ns["Prompt"](e, w, stop)(xr)
        (gr,) = torch.autograd.grad(loss, xr)
        rec[f"w_{tag}"] = np.float32(w); rec[f"stop_{tag}"] = np.float32(stop)
        rec[f"loss_{tag}"] = loss.detach().numpy(); rec[f"grad_{tag}"] = gr.numpy()
    rec["sdl"] = ns["spherical_dist_loss"](x, e[:1].expand(8, -1)).numpy()
    np.savez(os.path.join(out, "prompt_golden.npz"), **rec)

    vs = rx.vqgan_ns()
    xq = torch.randn(1, 4, 4, 16, generator=g)
    cb = torch.randn(64, 16, generator=g)
    xr = xq.clone().requires_grad_(True)
    q = vs["vector_quantize"](xr, cb)
    gq = torch.randn(1, 4, 4, 16, generator=g)
    (gx,) = torch.autograd.grad(q, xr, gq)
    u = torch.randn(2, 3, 8, 8, generator=g) * 0.8 + 0.5
    ur = u.clone().requires_grad_(True)
    c = vs["clamp_with_grad"](ur, 0, 1)
    gc = torch.randn(2, 3, 8, 8, generator=g)
    (gu,) = torch.autograd.grad(c, ur, gc)
    np.savez(os.path.join(out, "vq_clamp_golden.npz"), x=xq.numpy(), codebook=cb.numpy(), q=q.detach().numpy(),
             gq=gq.numpy(), gx=gx.numpy(), u=u.numpy(), c=c.detach().numpy(), gc=gc.numpy(), gu=gu.numpy())

    # ---- independent implementations of the un-vendored towers ---------------------------------------------------
    p = weights.synthetic_clip_vit_params(GOLDEN_CLIP, 21)
    m = hf_clip_from_params(GOLDEN_CLIP, p)
    xin = torch.randn(3, 3, 64, 64, generator=g)
    xr = xin.clone().requires_grad_(True)
    emb = m(pixel_values=xr).image_embeds
    ge = torch.randn(3, 64, generator=g)
    (gxi,) = torch.autograd.grad(emb, xr, ge)
    np.savez(os.path.join(out, "clip_vit_golden.npz"), x=xin.numpy(), emb=emb.detach().numpy(), ge=ge.numpy(),
             gx=gxi.numpy(), seed=np.int64(21))

    pv = weights.synthetic_vqgan_params(GOLDEN_VQ, 22)
    d = hf_decoder_from_params(GOLDEN_VQ, pv)
    zq = torch.randn(1, 128, 8, 8, generator=g)
    zr = zq.clone().requires_grad_(True)
    img = d(zr)
    gi = torch.randn(1, 3, 16, 16, generator=g)
    (gz,) = torch.autograd.grad(img, zr, gi)
    np.savez(os.path.join(out, "decoder_golden.npz"), z=zq.numpy(), img=img.detach().numpy(), gi=gi.numpy(),
             gz=gz.numpy(), seed=np.int64(22))
    pe = weights.synthetic_vqgan_encoder_params(GOLDEN_VQ, 23)
    enc = hf_encoder_from_params(GOLDEN_VQ, pe)
    xi = torch.rand(1, 3, 16, 16, generator=g) * 2 - 1
    with torch.no_grad():
        hz = enc(xi)
    np.savez(os.path.join(out, "encoder_golden.npz"), x=xi.numpy(), h=hz.numpy(), seed=np.int64(23))
    pt = weights.synthetic_clip_text_params(GOLDEN_TEXT, 24)
    mt = hf_clip_text_from_params(GOLDEN_TEXT, pt)
    tk = golden_tokens(GOLDEN_TEXT, 5, g)
    with torch.no_grad():
        te = mt(input_ids=tk).text_embeds
    np.savez(os.path.join(out, "clip_text_golden.npz"), tokens=tk.numpy(), emb=te.numpy(), seed=np.int64(24))
    for f in sorted(os.listdir(out)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(out, f)), "bytes")


if __name__ == "__main__":
    main()
