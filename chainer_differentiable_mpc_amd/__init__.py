"""placeholder - filled in below"""
