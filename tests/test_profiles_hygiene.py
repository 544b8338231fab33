"""Evidence files under profiles/ are what their names say: no rNN_* file is a byte copy of another round's file of the
same name (tools/check_profiles.py)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_round_evidence_file_is_a_copy_of_another_rounds():
    spec = importlib.util.spec_from_file_location("check_profiles", os.path.join(ROOT, "tools", "check_profiles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.duplicates() == []
