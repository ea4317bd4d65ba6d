"""Drop-in for the reference's `lvdm` package (third_party/ViewCrafter/lvdm): the module paths the guidedvd drivers import
(`utils_vc/diffusion_utils.py:8-10`) and the dotted `target:` strings of `configs/inference_pvd_{512,1024}.yaml` resolve
here when `guidedvd-3dgs_amd/` precedes `third_party/ViewCrafter` on `sys.path` (the reference appends the latter,
utils/viewcrafter_wrapper.py:26).  Every name is a thin alias of the MI355X-native implementation in `lvdm_amd`."""
