"""Drop-in for the reference's ``simple_knn`` extension package (``.gitmodules:1-3``; imported
as ``from simple_knn._C import distCUDA2`` at ``scene/gaussian_model.py:20``)."""
