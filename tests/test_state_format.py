"""Checkpoint wire format against a file written by the reference's own save_state (output/state.py)."""
import os

import numpy as np
import torch

from tests.util import GOLDEN


def test_file_written_by_the_reference_is_read_and_reproduced(tmp_path):
    from wxfactory_amd.state import load_state, save_state

    g = np.load(os.path.join(GOLDEN, "state_file_v.npz"))
    ref_file = tmp_path / "ref.npy"
    ref_file.write_bytes(g["file_bytes"].tobytes())
    state, version, config = load_state(str(ref_file))
    assert np.array_equal(state, g["state"]) and version == str(g["state_version"])
    assert config == str(g["config_text"]).strip()
    mine = tmp_path / "mine.npy"
    save_state(torch.from_numpy(g["state"]), version, str(g["config_text"]), str(mine))
    assert mine.read_bytes() == ref_file.read_bytes()  # byte-identical to the reference's writer


def test_global_layout_round_trip_single_rank():
    from wxfactory_amd.state import distribute_cube, gather_cube

    x = torch.arange(6 * 3 * 4 * 4 * 9, dtype=torch.float64).reshape(6, 3, 4, 4, 9)
    assert torch.equal(distribute_cube(gather_cube(x)), x)
