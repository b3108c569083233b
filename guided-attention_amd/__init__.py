"""guided-attention_amd — the per-step guided-attention hot path of jackBonadies/Guided-Attention,
built from scratch for AMD MI355X (gfx950): hand-written HIP kernels behind a C ABI
(`libga_hip.so`, see include/ga_hip.h) and a Python host layer that mirrors the reference's
own interface for this path (processor / AttentionStore / pipeline / RunConfig / run).

Import as `guided_attention_amd` (alias package at the repo root).
"""
__version__ = "0.1.0"
