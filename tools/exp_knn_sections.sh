# dev helper: builds lib/libpcc_nn_sect.so = the library with tools/exp_knn_sections.patch applied (wave time per section of
# k_grid_knn_sel), then restores the product source and library.  Run from the repo root, on the build host (no GPU needed).
set -e
git apply tools/exp_knn_sections.patch
trap 'git apply -R tools/exp_knn_sections.patch; make lib > /dev/null' EXIT
make lib > /dev/null
cp pointcloudcomparator_amd/lib/libpcc_nn.so pointcloudcomparator_amd/lib/libpcc_nn_sect.so
