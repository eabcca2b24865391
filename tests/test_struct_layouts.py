"""The ctypes mirrors of the C-ABI structs must have the layout the C compiler gives include/deeprl_amd.h: size and the offset
of every field, checked by compiling a probe with gcc (no GPU, no HIP)."""
import ctypes
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_layout(tmp_path, struct, fields):
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "include/deeprl_amd.h"', 'int main(void) {',
           '  printf("%%zu\\n", sizeof(%s));' % struct]
    src += ['  printf("%%zu\\n", offsetof(%s, %s));' % (struct, f) for f in fields]
    src += ['  return 0;', '}']
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "probe")
    subprocess.run(["gcc", "-std=c99", str(c), "-I", ROOT, "-o", exe], check=True)
    return [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
@pytest.mark.parametrize("which", ["dra_per_chain2_io", "dra_dqn_step_params", "dra_fold_seg", "dra_ppo_mlp_net", "dra_ppo_mlp_cfg",
                                   "dra_ppo_mlp_rollout_io"])
def test_ctypes_mirror_matches_the_header(tmp_path, which):
    from deeprl_amd import ops
    from deeprl_amd.learner import StepParams
    from deeprl_amd import ppo_mlp
    mirror = {"dra_per_chain2_io": ops.PerChain2IO, "dra_dqn_step_params": StepParams, "dra_fold_seg": ops.FoldSeg,
              "dra_ppo_mlp_net": ppo_mlp.Net, "dra_ppo_mlp_cfg": ppo_mlp.Cfg, "dra_ppo_mlp_rollout_io": ppo_mlp.RolloutIO}[which]
    names = [f[0] for f in mirror._fields_]
    got = _c_layout(tmp_path, which, names)
    assert got[0] == ctypes.sizeof(mirror)
    assert got[1:] == [getattr(mirror, n).offset for n in names]
