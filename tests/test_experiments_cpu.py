"""CPU: every patch under experiments/ (code that was built, measured and not adopted: experiments/README.md) names its base commit and
still applies to it -- `git apply --check` against a scratch index of that commit, the working tree untouched.  Skipped where the history
is not available (the GPU box receives the tree without .git)."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "experiments", "*.patch")))


def _git(*args, env=None):
    return subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True, env=env)


def test_every_experiment_is_listed_in_the_readme():
    readme = open(os.path.join(ROOT, "experiments", "README.md")).read()
    assert PATCHES, "experiments/ holds no patch"
    for p in PATCHES:
        assert f"`{os.path.basename(p)}`" in readme, f"{os.path.basename(p)} is not described in experiments/README.md"


@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_patch_applies_to_its_base_commit(patch, tmp_path):
    head = open(patch).read(4000)
    m = re.search(r"^# base: ([0-9a-f]{7,40})\s*$", head, re.M)
    assert m, "no '# base: <commit>' header"
    for key in ("# what:", "# measured:", "# reproduce:"):
        assert key in head, f"no '{key}' header"
    if not os.path.isdir(os.path.join(ROOT, ".git")) or _git("cat-file", "-e", m.group(1) + "^{commit}").returncode != 0:
        pytest.skip("no git history here (or the base commit is not in it)")
    env = dict(os.environ, GIT_INDEX_FILE=str(tmp_path / "index"))
    r = _git("read-tree", m.group(1), env=env)
    assert r.returncode == 0, r.stderr
    r = _git("apply", "--cached", "--check", patch, env=env)
    assert r.returncode == 0, f"{os.path.basename(patch)} does not apply to {m.group(1)}:\n{r.stderr}"
