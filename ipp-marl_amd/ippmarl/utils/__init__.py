"""ippmarl.utils: drop-in counterparts of the reference package of the same name."""
