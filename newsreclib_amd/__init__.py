"""newsreclib_amd -- MI355X (gfx950) native NRMS hot path behind NewsRecLib's module interfaces.

Public surface (mirrors the reference's operator API for this path):

* ``newsreclib_amd.nrms_module.NRMSModule``  -- drop-in ``model._target_``
* ``newsreclib_amd.news_encoder.{MHSAAddAtt, NewsEncoder}``, ``user_encoder.UserEncoder``,
  ``click_predictor.DotProduct`` -- the sub-module interfaces
* ``newsreclib_amd.trainer.NRMSTrainer`` -- flat-buffer train step with RCCL data parallelism
* the C ABI itself: ``include/newsreclib_amd.h`` / ``newsreclib_amd/libnewsreclib_amd.so``
"""
__version__ = "0.1.0"
