"""Alias of wgsassign_amd.WGSassign: `python -m WGSassign.WGSassign` / console script `WGSassign`."""
from wgsassign_amd.WGSassign import main, parser  # noqa: F401

if __name__ == "__main__":
    main()
