#!/usr/bin/env python3
"""Replace the generated measurement block of DESIGN.md (between the profiles:begin / profiles:end markers) with what
tools/summarize_profiles.py --markdown <tag> prints today.  Usage: python tools/refresh_design_block.py r04"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
block = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), "--markdown", tag], stdout=subprocess.PIPE,
                       text=True, check=True).stdout
path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
new = re.sub(r"<!-- profiles:begin \w+ -->\n.*?<!-- profiles:end -->", lambda m: f"<!-- profiles:begin {tag} -->\n{block}<!-- profiles:end -->",
             text, flags=re.S)
assert new != text or block in text
open(path, "w").write(new)
print(f"DESIGN.md: block of {tag} refreshed ({len(block)} characters)")
