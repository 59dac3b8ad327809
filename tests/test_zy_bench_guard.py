"""bench.py marks the counter-derived keys of its roofline block stale when the code the library is built from no longer
matches the code the committed counter passes were taken from. The hash covers code and compiler flags, not comments."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_hash_ignores_comments_and_layout_but_not_code():
    import bench
    a = 'int f(int x) { return x + 1; }  // adds one\n/* block\n comment */ const char* s = "a // b /* c */";\n'
    b = 'int f(int x)\n{\n    return x + 1;   /* different words */\n}\nconst char* s = "a // b /* c */";   // trailing\n'
    assert bench._code_only(a) == bench._code_only(b)
    assert bench._code_only(a) != bench._code_only(a.replace("x + 1", "x + 2"))
    assert bench._code_only(a) != bench._code_only(a.replace('"a // b /* c */"', '"a // b"'))      # literals are code
    assert "'\"'" in bench._code_only("char q = '\"'; // quote")


def test_committed_counters_carry_the_hash_of_the_committed_sources():
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16
    for name in ("r3_roofline_counters.json", "r3_roofline_counters_interior.json"):
        prof = json.loads((ROOT / "profiles" / name).read_text())
        assert prof["kernel_source_hash"] == h, f"profiles/{name} was taken from other kernel sources: re-run tools/profile_r3.sh"
