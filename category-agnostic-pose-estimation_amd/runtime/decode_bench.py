"""Measurement of the KV-cached autoregressive decode (BASELINE.json configs[4]; SURVEY 8d: images/s and time per decode
step): synthetic K-shot episodes through `CAPEModel.forward_inference`, the decode loop timed by HIP events on the stream
it runs on (image encoding excluded), eagerly and as replayed per-step hipGraphs.  Used by bench.py and tools/decode_bench.py."""
import argparse
import os

import torch


def decoder_step_bytes(model):
    """Bytes of decoder weights one decode step streams (fp32, every tensor once): folded q|k|v projection, self-attention
    out_proj, support-attention q and out projections, sampling_offsets|attention_weights, output_proj, FFN, norms, coords MLP
    per layer, plus pos_trans(+norm) for layers 1.., the last class head and the token-embedding rows touched (4 per image)."""
    dec = model.base_model.transformer.decoder
    n = 0
    for li, l in enumerate(dec.layers):
        n += 768 * 256 + 256 * 256 + 768                                  # folded qkv + in_proj_q (query position) + bias
        n += sum(p.numel() for p in l.self_attn.out_proj.parameters())
        n += 256 * 256 + 256 + sum(p.numel() for p in l.support_attn.out_proj.parameters())
        m = l.cross_attn
        n += sum(p.numel() for q in (m.sampling_offsets, m.attention_weights, m.output_proj) for p in q.parameters())
        n += sum(p.numel() for q in (l.linear1, l.linear2, l.norm1, l.norm2, l.norm3, l.norm_support) for p in q.parameters())
        n += sum(p.numel() for p in dec.coords_embed[li].parameters())
        if li + 1 < len(dec.layers):
            n += sum(p.numel() for q in (dec.pos_trans, dec.pos_trans_norm) for p in q.parameters())
    n += sum(p.numel() for p in dec.class_embed[-1].parameters())
    return 4 * n


def decode_benchmark(device, episodes=1, image_size=512, keypoints=68, shots=5, queries=2, reps=3, seed=3, model=None, tok=None):
    import cape_amd  # noqa: F401
    from cape_amd.datasets import DiscreteTokenizerV2, episodic_collate_fn
    from cape_amd.datasets.synthetic import SyntheticEpisodes
    from cape_amd.models import build_model
    from cape_amd.models.cape_model import build_cape_model
    from cape_amd.models.train_cape_episodic import get_args_parser
    if model is None:
        args = argparse.ArgumentParser(parents=[get_args_parser()]).parse_args(
            ["--use_geometric_encoder", "--use_gcn_preenc", "--image_size", str(image_size)])
        torch.manual_seed(0)
        tok = DiscreteTokenizerV2(44, args.seq_len)
        base, _ = build_model(args, tokenizer=tok)
        model = build_cape_model(args, base).to(device).eval()
        with torch.no_grad():                                   # random-init heads say <eos> at once: keep the loop running for
            model.base_model.class_embed[-1].bias.copy_(torch.tensor([4.0, 0.0, -4.0]))     # the full 200 steps (all <coord>)
    ds = SyntheticEpisodes(tok, episodes, image_size, keypoints, queries, num_support=shots, seed=seed)
    b = episodic_collate_fn([ds[j] for j in range(episodes)])
    im, sc, sm, sk = b["query_images"].to(device), b["support_coords"].to(device), b["support_masks"].to(device), b["support_skeletons"]
    os.environ["WARN_INCOMPLETE_GENERATION"] = "0"
    N = im.shape[0]

    def run(graph):
        out = model.forward_inference(im, sc, sm, skeleton_edges=sk, graph=graph, timing=True)
        torch.cuda.synchronize()
        tm = out["_timing"]
        whole[0] = bool(tm.get("whole_step_kernel"))
        return tm["events"][0].elapsed_time(tm["events"][1]), tm["steps_run"], tm["launch"], tm["fused"]

    res = {}
    whole = [False]
    with torch.no_grad():
        run(False)                                              # warm-up (allocator, lazily set kernel attributes)
        ms, steps, _, fused = min(run(False) for _ in range(reps))
        res["eager"] = {"ms_per_step": ms / steps, "steps": steps}
        run(True); run(True)                                    # call 2 of the geometry captures the per-step graphs, then replays
        ms, steps, launch, fused = min(run(True) for _ in range(reps))
        res["graph"] = {"ms_per_step": ms / steps, "steps": steps, "launch": launch}
    nbytes = decoder_step_bytes(model)
    best = min(res["eager"]["ms_per_step"], res["graph"]["ms_per_step"])
    return {"workload": f"{shots}-shot KV-cached decode, {image_size}x{image_size}, {keypoints} support keypoints, "
                        f"{episodes} episode(s) x {queries} queries = {N} images in flight",
            "images": N, "image_size": image_size, "support_keypoints": keypoints, "shots": shots, "steps": res["graph"]["steps"],
            "fused_step_kernels": bool(fused), "whole_step_kernel": whole[0], "launches_per_step": 2 if whole[0] else (75 if fused else 190),
            "us_per_step_eager": round(res["eager"]["ms_per_step"] * 1e3, 1), "us_per_step_graph": round(res["graph"]["ms_per_step"] * 1e3, 1),
            "tokens_per_s": round(N / (best * 1e-3), 1), "images_per_s_200_steps": round(N / (best * 200 * 1e-3), 2),
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0, "achieved": round(nbytes / (best * 1e-3) / 1e9, 2),
                         "frac": round(nbytes / (best * 1e-3) / 1e9 / 8000.0, 5), "algorithmic_bytes_per_step": nbytes,
                         "note": "decoder weights counted once per step for the whole batch (SURVEY 8d); the step is a chain of "
                                 "dependent small products, i.e. latency bound, not HBM bound"},
            # the whole-step kernel streams every decoder weight through EACH image's compute unit (one block per image):
            # its own bound is what one CU can ingest from L2 / Infinity Cache (tools/lab/cu_ingest.hip: 107-120 GB/s)
            "cu_stream": {"bound": "per-CU ingest", "unit": "GB/s", "peak": 110.0, "achieved": round(nbytes / (best * 1e-3) / 1e9, 2),
                          "frac": round(nbytes / (best * 1e-3) / 1e9 / 110.0, 4)} if whole[0] else None}
