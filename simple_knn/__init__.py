"""Drop-in module name for the reference's `from simple_knn._C import distCUDA2`
(/root/reference/scene/gaussian_model.py:21).  Implementation: egogaussian_amd (HIP, gfx950)."""
