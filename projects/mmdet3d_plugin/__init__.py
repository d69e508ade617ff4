"""`plugin_dir='projects/mmdet3d_plugin/'` of the shipped configs resolves here (ref: extra_tools/train.py:105-127):
importing it registers the MI355X-native implementations under the reference's registry names."""
from uni3detr_amd.plugin import *  # noqa: F401,F403
