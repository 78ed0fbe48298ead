"""CPU oracle for the kod YOLOv5 training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch / numpy CPU
restatement of the reference algorithm (craston/object_detection_cib, package
``kod``).  It exists so that the hand-written HIP path can be checked against
it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it; the product package
``object_detection_cib_amd`` never does (``tests/test_no_oracle_leak.py``
enforces that).

Parity status: PINNED.  Every function below is checked against golden
vectors produced by importing the real reference in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).  Pieces whose arithmetic
lives in third-party wheels that are absent from the reference tree and from
this image (OpenCV ``warpAffine``/``cvtColor``, pycocotools ``COCOeval``) are
marked "parity unpinned" in their own module headers.

Each function cites the reference file:line it follows (paths relative to the
reference checkout root).
"""
