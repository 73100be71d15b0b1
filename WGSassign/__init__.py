"""Import-path drop-in: `from WGSassign import emMAF, glassy, emMAF_cy, glassy_cy, reader_cy, utils,
fisher` and `python -m WGSassign.WGSassign` resolve to the MI355X implementation in
`wgsassign_amd` (the reference's package has these module names: setup.py:47, WGSassign.py:148-159)."""
