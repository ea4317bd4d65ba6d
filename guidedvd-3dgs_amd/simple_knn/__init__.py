"""MI355X-native drop-in for the reference's `simple_knn` package (submodules/simple-knn): `from simple_knn._C import distCUDA2`."""
