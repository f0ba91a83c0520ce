"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2)
into the per-kernel table that --stats would print: calls, total / mean / min / max duration."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                            f"max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) from kernels group by {name_col} order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = ['%-86s %6s %12s %10s %10s %10s %6s %5s %5s %7s %7s' % ('kernel', 'calls', 'total_ms', 'mean_us', 'min_us', 'max_us', '%', 'vgpr', 'agpr', 'lds', 'scratch')]
    for r in rows:
        lines.append('%-86s %6d %12.3f %10.1f %10.1f %10.1f %6.2f %5s %5s %7s %7s' % (r[0][:86], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
                                                                               100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
