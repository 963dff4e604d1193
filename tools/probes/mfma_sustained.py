"""What does the chip sustain on FP64 MFMAs ALONE (register operands, no LDS, no memory), for as long as a Gram takes?
rt_bench_mfma_f64 with 4 / 8 / 16 accumulators per wave, 1 / 2 / 4 waves per SIMD, launches of 5-40 ms.
The Gram kernel's 57-62 TF is to be read against THIS number, not against 78.6 (2.4 GHz x 32 flop/clk/SIMD)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from romtime_amd import ops
for nacc_log2 in (2, 3, 4):
    for wg in (1, 2, 4):
        it = 400_000 >> nacc_log2
        tfs = [ops.bench_mfma_f64(it | (nacc_log2 << 24) | (wg << 28)) for _ in range(3)]
        ms = 256 * wg * 4 * it * (1 << nacc_log2) * 2048.0 / (tfs[-1] * 1e12) * 1e3
        print(f"accumulators/wave={1 << nacc_log2} waves/SIMD={wg}: {tfs[0]:.1f} {tfs[1]:.1f} {tfs[2]:.1f} TF  (~{ms:.1f} ms per launch)", flush=True)
