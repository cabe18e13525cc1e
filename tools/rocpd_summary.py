#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace stats and/or PMC counters) as text.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [more.db ...] > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print("== %s" % path)
        try:
            rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
            print("-- kernel stats (durations in us)")
            print("%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
            for name, calls, tot, avg, pct in rows:
                print("%-110s %8d %14.1f %12.2f %7.2f" % (name[:110], calls, tot, avg, pct))
        except sqlite3.Error as e:
            print("(no top_kernels: %s)" % e)
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                "from counters_collection group by kernel_name, counter_name"))
            if rows:
                print("-- PMC counters (per dispatch)")
                print("%-90s %-14s %6s %16s %16s %16s %12s" % ("kernel", "counter", "n", "avg", "min", "max", "avg_ns"))
                for k, c, n, a, lo, hi, d in rows:
                    print("%-90s %-14s %6d %16.2f %16.2f %16.2f %12.0f" % (k[:90], c, n, a, lo, hi, d))
        except sqlite3.Error as e:
            print("(no counters: %s)" % e)


if __name__ == "__main__":
    main()
