"""INTEGRATION.md's "which layout takes which kernel" table is generated from the sources (tools/kernel_table.py: the lean kernel's
instantiation lists, the matrix kernel's admission rule): the document and the code may not part ways."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_carries_the_generated_table():
    table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py")], capture_output=True, text=True, check=True).stdout
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = doc[doc.index("<!-- kernel-table:begin -->") + len("<!-- kernel-table:begin -->"):doc.index("<!-- kernel-table:end -->")]
    assert body.strip() == table.strip(), "run `python tools/kernel_table.py` and paste its output between the kernel-table markers of INTEGRATION.md"
    assert table.count("src_lean_kernel<") == 44 and table.count("only block kernel") == 28
