"""CPU oracle for the rvc/infer hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (torch-CPU fp32 / numpy f64) of the reference algorithm, each
function citing the reference file:line it follows.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (polgen-rvc_amd/) never does.

Pinning status (see DESIGN.md "Oracle"):
  * synthesizer / RMVPE / pipeline arithmetic: PINNED against the reference's own modules
    imported in the build container (tools/gen_golden.py -> tests/golden/*.npz).
  * HuBERT (fairseq 0.12.2, not vendored in the reference) and FAISS (faiss-cpu 1.7.3):
    PARITY UNPINNED by the reference; cross-checked against transformers.HubertModel and
    exact float64 brute force respectively.
"""
