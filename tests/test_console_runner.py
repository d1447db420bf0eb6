"""The headless runner (python -m raytracer_amd): scene pick by name, RenderOptions from arguments, PNG save."""
import numpy as np
import pytest

from raytracer_amd import __main__ as runner
from raytracer_amd import builtin_scenes as bs


def test_scene_table_is_get_all_plus_the_benchmark_scenes(capsys):
    assert runner.main(["--list"]) == 0
    names = capsys.readouterr().out.split("\n")[:-1]
    assert names[:6] == [s.Name for s in bs.GetAll()] and len(names) == 8


def test_unknown_scene_and_visualisation_are_usage_errors(capsys):
    assert runner.main(["--scene", "no such scene"]) == 2
    assert runner.main(["--scene", "Demo", "--vis", "Pretty"]) == 2
    assert runner.main(["--scene", "e"]) == 2            # ambiguous
    assert "ambiguous" in capsys.readouterr().err


def test_visualisation_names_follow_the_enum():
    assert runner.VIS_NAMES["None"] == 0 and runner.VIS_NAMES["Normals"] == 1 and runner.VIS_NAMES["BounceDepth"] == 13


@pytest.mark.gpu
def test_runner_saves_the_job_image(tmp_path):
    from PIL import Image
    import raytracer_amd as rt
    from raytracer_amd.core import RenderOptions
    out = tmp_path / "image.png"
    assert runner.main(["--scene", "Cornell Box", "--width", "96", "--height", "96", "--passes", "6", "--depth", "12", "--seed", "0x77", "--out", str(out),
                        "--progress-passes", "2"]) == 0
    png = np.asarray(Image.open(out))
    job = rt.RenderJob(bs.CornellBox(), RenderOptions(RenderWidth=96, RenderHeight=96, Passes=6, MaxBounceDepth=12, Seed=0x77)); job.Wait()
    assert png.shape == (96, 96, 3) and np.array_equal(png, job.Image) and png.max() > 0
    job.Dispose()
    assert runner.main(["--scene", "Demo", "--width", "64", "--height", "36", "--passes", "1", "--vis", "Normals", "--out", str(out)]) == 0
    assert np.asarray(Image.open(out)).shape == (36, 64, 3)


@pytest.mark.gpu
def test_runner_renders_an_obj_file(tmp_path):
    from PIL import Image
    obj = tmp_path / "quad.obj"; obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    out = tmp_path / "o.png"
    assert runner.main(["--obj", str(obj), "--width", "64", "--height", "36", "--passes", "2", "--out", str(out)]) == 0
    assert np.asarray(Image.open(out)).shape == (36, 64, 3)
