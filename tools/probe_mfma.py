from romtime_amd import ops
for lacc in (1, 2, 3, 4):
    for wg in (1, 2, 4):
        it = 20000 // (1 << lacc) * 4
        tf = ops.bench_mfma_f64(it | (lacc << 24) | (wg << 28))
        print(f"acc={1<<lacc} wg/cu={wg} waves/simd={wg}: {tf:.1f} TF")
