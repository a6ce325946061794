"""Turn a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table that
`rocprofv3 --kernel-trace --stats` prints (name, calls, total/avg/min/max ns, share)."""
import sqlite3
import sys


def summarize(db_path, out_path=None, top=80):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = cur.execute(
        "select k.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol k on d.kernel_id = k.id "
        "group by k.kernel_name order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["name,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for r in rows[:top]:
        name = r[0].replace(",", ";")
        lines.append("%s,%d,%d,%.1f,%d,%d,%.2f" % (name[:120], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
    text = "\n".join(lines) + "\n"
    if out_path:
        with open(out_path, "w") as f:
            f.write(text)
    return text


if __name__ == "__main__":
    sys.stdout.write(summarize(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))
