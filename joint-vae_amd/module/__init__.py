"""Drop-in `module` package of the MI355X-native joint-CVAE (same import surface as the reference's module/)."""
