// plane-fed bf16 convolutions: {M, N, C, taps, nphase, S, 0, ks},
// (empty: a per-shape table from serialised per-layer timings -- profiles/r05l_tune_bf16p_report.txt, 1.1 of 8.8 ms -- measured
//  -0.8 % on the concurrent iteration and is not shipped; tools/tune_x3.py sweep_bf16 / table_bf16 regenerate it)
