protein_letters_3to1 = {}
