python tools/probe/epi_variants.py 53248 2 2>&1 | grep -v amdgpu.ids
