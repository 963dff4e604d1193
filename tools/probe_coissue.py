from romtime_amd import ops
it = 200000
for mode, name in [(1, "mfma only"), (2, "dp fma only"), (3, "mfma + dp fma"), (4, "int valu only"), (5, "mfma + int valu")]:
    ms = ops.bench_mfma_f64(it | (15 << 24) | (mode << 28))
    print(f"mode {mode} {name}: {ms:.3f} ms", flush=True)
