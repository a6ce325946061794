"""The kernels of the fast parse, the post stage and the validity gate under the race check (tests/race): the host emulation with
a launch's wavefronts on eight host threads, built with ThreadSanitizer.  Rule enforced: a thread acts only on state written
by an EARLIER launch (or through the atomics / the two reads and stores the source names as meant to race) -- the rule whose
violation by FastWordCheck wrote round 3's undecodable members.  Evidence that the check sees that defect in the old kernel:
profiles/r04_race_selftest_17MiB.log (`tests/race/run.sh -s`, 40 minutes: too long for this tier); here a small mixed input
that takes the repair passes, the redo with finer tiles and the tail stage."""
import os
import subprocess

import _data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_kernel_reads_what_another_wavefront_of_its_launch_writes(tmp_path, oracle):
    data = _data.mixed(200_000, seed=3)
    src = tmp_path / "mixed.bin"
    src.write_bytes(data)
    env = dict(os.environ, RACE_OUT=os.path.join(ROOT, "build", "race"))
    r = subprocess.run([os.path.join(ROOT, "tests", "race", "run.sh"), "fast", str(src)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "0 data races reported" in r.stdout and "rc 0" in r.stdout, r.stdout[-3000:]
