"""End-to-end check of the tgsfilter command line against a golden case produced by the reference:
output file byte-equal, stderr INFO lines equal, HTML <tr> rows and `var data` object equal."""
from __future__ import annotations

import gzip
import json
import os
import re
import subprocess
import tempfile


def run_case(binary: str, golden_dir: str, name: str, extra_args=()):
    cmd = json.load(open(os.path.join(golden_dir, name + ".cmd.json")))
    ref_out = gzip.open(os.path.join(golden_dir, name + ".out.fq.gz"), "rb").read()
    ref_err = open(os.path.join(golden_dir, name + ".stderr.txt")).read()
    ref_html = json.load(open(os.path.join(golden_dir, name + ".html.json")))
    with tempfile.TemporaryDirectory() as td:
        fmt = cmd.get("in_format", "fq")
        fin = os.path.join(td, "in." + fmt)
        if fmt == "bam":
            open(fin, "wb").write(open(os.path.join(golden_dir, name + ".in.bam"), "rb").read())
        else:
            open(fin, "wb").write(gzip.open(os.path.join(golden_dir, name + ".in." + fmt + ".gz"), "rb").read())
        args = [binary, "-i", fin, "-t", "1"] + cmd["flags"].split() + list(extra_args)
        qc = "--qc" in cmd["flags"]
        if not qc:
            args += ["-o", os.path.join(td, "out.fa" if fmt == "fa" else "out.fq")]
        if cmd["adapters"]:
            fa = os.path.join(td, "adapters.fa")
            with open(fa, "w") as f:
                for i, a in enumerate(cmd["adapters"]):
                    f.write(">a%d\n%s\n" % (i, a))
            args += ["-a", fa]
        p = subprocess.run(args, capture_output=True, cwd=td)
        err = p.stderr.decode().replace(td + "/", "")
        assert p.returncode == 0, err
        out = open(os.path.join(td, "out.fa" if fmt == "fa" else "out.fq"), "rb").read() if not qc else b""
        html = open(os.path.join(td, "in.html" if qc else "out.html"), encoding="utf-8").read()
    assert out == ref_out, "output FASTQ differs from the reference's"

    def info(text):
        lines = [l for l in text.splitlines() if l.startswith("INFO:") or l.startswith("Warning:")]
        # the set of -a adapters is an unordered_set in the reference: compare it as a set
        ad = sorted(l.split(":", 2)[2] for l in lines if l.startswith("INFO: input adapter"))
        rest = [l for l in lines if not l.startswith("INFO: input adapter") and not l.startswith("Warning: reset -t")]
        return ad, rest
    assert info(err) == info(ref_err), "stderr differs:\n%s\n---- reference:\n%s" % (err, ref_err)
    rows = re.findall(r"<tr>.*?</tr>", html, flags=re.S)
    assert rows == ref_html["table_rows"], (rows, ref_html["table_rows"])
    m = re.search(r"var data = (\{.*?\})\n</script>", html, flags=re.S)
    got = m.group(1) if m else None
    ref = ref_html["data"]
    assert got is not None and ref is not None
    assert got.strip() == ref.strip(), _first_diff(got, ref)


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return "var data differs at %d: ...%r vs ...%r" % (i, a[max(0, i - 60):i + 60], b[max(0, i - 60):i + 60])
    return "var data lengths differ: %d vs %d" % (len(a), len(b))
