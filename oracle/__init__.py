"""CPU oracle for the RGBD-GAN hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in NumPy and torch-CPU fp32, the arithmetic that the
reference (nogu-atsu/RGBD-GAN, Chainer/CuPy) performs on the path
``RGBDUpdater.update_core`` (updater.py:274-448) -> generator / discriminator
(net.py) -> ``LossFuncRotate`` (common/loss_functions.py:31-228), and on the
DeepVoxels variant ``DeepVoxelsUpdater.update_core`` (updater_deepvoxels.py:123-252)
-> ``deepvoxels_generator.Generator`` -> deepvoxel/projection.py, deepvoxel/deepvoxel.py.
Each function cites the reference file:line it follows.

* It is the checker, never the product: only ``tests/``,
  ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
  import it.  Nothing under ``rgbd_gan_amd/`` imports it, and the product path
  raises if the HIP library is missing instead of falling back to this code.
* PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures,
  and Chainer/CuPy cannot be installed in the build container, so the oracle
  cannot be checked against outputs of the reference itself.  It is pinned to
  the known-answer tests listed in SURVEY.md section 8(c) (tests/test_oracle_*.py)
  and to fixtures it generated itself (tests/golden/, script committed).
  The Chainer semantics it assumes are listed in DESIGN.md.
"""
