#!/usr/bin/env python3
"""The document around the report's numbers, as DATA taken from reports the reference itself writes.

north_star keeps "the HTML/report.cpp surface unchanged": a report of this build must be the reference's document
(include/report.cpp:670-700: header, style, the plot containers of the six report kinds, the chart code = Apache
ECharts (Apache-2.0) + changeDPI (MIT) as the reference embeds them, include/echart_js.cpp, and its own glue script)
with this run's table rows, plot data and time stamp in it.  This script RUNS the reference
(oracle/_ref/tgsfilter_ref, built by oracle/Makefile from the sources where they lie) once per report kind, cuts each
document at fixed markers and stores the pieces under tgsfilter_amd/host/assets/:
    report_head.html        everything before <table>
    report_plots_<qc>.html  between </table> and the footer, per report kind (qc = 00 01 02 10 11 12, :3285-3299)
    report_charts.js        from </body> to the data script (the embedded chart libraries)
    report_tail.html        after the data script
The command line (tgsfilter_amd/host/report.cpp) puts table, footer and data between them.  Build container only."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tgsfilter_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "tgsfilter_ref")
OUT = os.path.join(ROOT, "tgsfilter_amd", "host", "assets")


def pieces(html):
    t0 = html.index("<table>")
    t1 = html.index("</table>\n") + len("</table>\n")
    f0 = html.index('<div id="footer">')
    b0 = html.index("</body>\n")
    d0 = html.index("<script>\nvar data = {")
    d1 = html.index("}\n</script>\n", d0) + len("}\n</script>\n")
    return html[:t0], html[t1:f0], html[b0:d0], html[d1:]


def main():
    os.makedirs(OUT, exist_ok=True)
    reads = synth.make_reads(3, 40, "ont", mean_len=3000)
    with tempfile.TemporaryDirectory() as td:
        synth.write_fastq(os.path.join(td, "in.fq"), reads)
        with open(os.path.join(td, "in.fa"), "wb") as f:
            for n, s, _ in reads:
                f.write(b">" + n + b"\n" + s + b"\n")
        runs = {"12": ["-i", "in.fq", "-o", "o.fq", "-q", "7"], "10": ["-i", "in.fq", "--qc"],
                "11": ["-i", "in.fq", "-o", "o.fq", "-F", "-r", "10"], "02": ["-i", "in.fa", "-o", "o.fa"],
                "00": ["-i", "in.fa", "--qc"], "01": ["-i", "in.fa", "-o", "o.fa", "-F", "-r", "10"]}
        common = None
        for qc, args in runs.items():
            p = subprocess.run([REF, "-x", "ont", "-t", "1"] + args, capture_output=True, cwd=td)
            assert p.returncode == 0, p.stderr.decode()
            name = [l for l in p.stderr.decode().splitlines() if "report was written to" in l][0].split("to: ")[1].rstrip(".")
            head, plots, charts, tail = pieces(open(os.path.join(td, name), encoding="utf-8").read())
            if common is None:
                common = (head, charts, tail)
            assert common == (head, charts, tail), "the document around the numbers differs between report kinds"
            open(os.path.join(OUT, "report_plots_%s.html" % qc), "w", encoding="utf-8").write(plots)
        open(os.path.join(OUT, "report_head.html"), "w", encoding="utf-8").write(common[0])
        open(os.path.join(OUT, "report_charts.js"), "w", encoding="utf-8").write(common[1])
        open(os.path.join(OUT, "report_tail.html"), "w", encoding="utf-8").write(common[2])
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
