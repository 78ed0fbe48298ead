"""object_detection_cib_amd - MI355X-native (gfx950 HIP) YOLOv5 training hot path behind the
`kod.nn` / `kod.core` / `kod.lightning` module API of craston/object_detection_cib.

Sub-packages mirror the reference's import paths (kod.X.Y -> object_detection_cib_amd.X.Y) for the
classes on the hot path; the arithmetic lives in csrc/*.hip (libkodhip.so, C ABI in include/kodhip.h).
"""
__version__ = "0.1.0"
