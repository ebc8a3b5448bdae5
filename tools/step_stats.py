"""Per-step kernel time by family from a rocprofv3 kernel_stats.csv of a bench.py run.
usage: python tools/step_stats.py KERNEL_STATS.csv [STEPS_TRACED]   (default: the number of adam_kernel launches, one per step)"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else float(sum(int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']))
print(f'{steps:.0f} steps traced')
fam = {}
FAMS = [('bn_bwd_apply', 'BN bwd apply'), ('bn_bwd_reduce', 'BN bwd reduce'), ('conv5_wgrad_x3', 'wgrad x3'), ('conv5_wgrad_kernel', 'wgrad f32'),
        ('wgrad_reduce', 'wgrad slab reduce'), ('conv5_x3_kernel', 'conv x3 (fwd/dgrad s1)'), ('convt2_x3', 'conv 4-phase x3'), ('convt2s_x3', 'conv 4-phase x3'), ('t2s_wpack', 'weight packs'),
        ('conv5_fwd_kernel', 'conv f32 mfma'), ('convt2_kernel', 'conv 4-phase f32'), ('gemm_kernel', 'dense products (gemm + split-K fold)'),
        ('gemm_x3_kernel', 'dense products (gemm + split-K fold)'), ('splitk_fold', 'dense products (gemm + split-K fold)'), ('small_transpose', 'unfold/fold'), ('pack_refresh', 'weight packs'), ('unfold', 'unfold/fold'),
        ('fold_kernel', 'unfold/fold'), ('smallco', '3-channel layers on the vector ALUs'), ('smallci', '3-channel layers on the vector ALUs'), ('sci_wpack', 'weight packs'), ('wpack', 'weight packs'), ('pack_kernel', 'weight packs'),
        ('bn_finalize', 'BN fwd'), ('bn_apply', 'BN fwd'), ('bn_stats', 'BN fwd'), ('adam', 'optimizer'), ('sqnorm', 'optimizer'),
        ('clip', 'optimizer')]
for r in rows:
    name = r['Name']
    key = next((v for k, v in FAMS if k in name), 'other')
    t = fam.setdefault(key, [0.0, 0])
    t[0] += float(r['TotalDurationNs']); t[1] += int(r['Calls'])
tot = sum(v[0] for v in fam.values())
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f'{k:26s} {v[0] / steps / 1e6:7.3f} ms/step  {v[1] / steps:6.1f} launches/step  {100 * v[0] / tot:5.1f} %')
print(f'{"total kernel time":26s} {tot / steps / 1e6:7.3f} ms/step')
print('-- top kernels')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:14]:
    nm = re.sub(r'\(anonymous namespace\)::', '', r['Name'])[:70]
    print(f'{nm:70s} calls/step {int(r["Calls"]) / steps:5.1f} avg {float(r["AverageNs"]) / 1e3:7.1f} us  {float(r["TotalDurationNs"]) / steps / 1e6:6.3f} ms/step')
